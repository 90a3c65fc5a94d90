set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/bf3_debug.py > gpurun_out/bf3_debug.txt 2>&1; grep -v "first bad" gpurun_out/bf3_debug.txt | tail -14
timeout 900 python -m pytest tests/test_gpu_snac.py tests/test_gpu_dac.py tests/test_gpu_encodec.py tests/test_gpu_soprano.py tests/test_gpu_qwen3tts.py tests/test_gpu_fullwidth.py -m gpu -q 2>&1 | grep -v "^PARITY" | tail -12
