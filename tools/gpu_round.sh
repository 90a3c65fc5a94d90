set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_soprano.py tests/test_gpu_sampler.py tests/test_gpu_snac.py tests/test_gpu_generate.py -m gpu -q 2>&1 | grep -v "^PARITY" | tail -15
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_samp2.log 2>&1; tail -1 gpurun_out/bench_samp2.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['phases_ms'], j['roofline']['step']['ms'])"
