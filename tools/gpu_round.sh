set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_qwen3tts.py tests/test_gpu_lm.py tests/test_gpu_generate.py tests/test_gpu_fullwidth.py -m gpu -q -x 2>&1 | tail -8
MIS_GEMM_NBUF=2 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_nbuf2.log 2>&1; tail -1 gpurun_out/bench_nbuf2.log | cut -c1-1800
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_nbuf3.log 2>&1; tail -1 gpurun_out/bench_nbuf3.log | cut -c1-1800
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_whisper -- python $GRAFT_REPO_ROOT/tools/bench_whisper.py 8 > $GRAFT_REPO_ROOT/gpurun_out/prof_whisper.log 2>&1)
find gpurun_out/prof_whisper -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -25 {}'
