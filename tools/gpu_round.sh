set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dac.py tests/test_gpu_qwen3tts.py tests/test_gpu_snac.py -m gpu -q -x 2>&1 | grep -v "^PARITY" | tail -25
timeout 300 python tools/bench_qwen3tts.py 32 100 16 > gpurun_out/q3_lowprio.log 2>&1; tail -3 gpurun_out/q3_lowprio.log
