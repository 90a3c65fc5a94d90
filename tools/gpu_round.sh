set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_lm.py -m gpu -q -x -rP 2>&1 | grep -v "^---" | tail -15
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
MIS_PREFILL_SEQ=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prefill_seq.log 2>&1; tail -1 gpurun_out/bench_prefill_seq.log | cut -c1-900
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prefill_batched.log 2>&1; tail -1 gpurun_out/bench_prefill_batched.log | cut -c1-900
