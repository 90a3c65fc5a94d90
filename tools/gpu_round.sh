set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_qwen3tts.py "tests/test_gpu_fullwidth.py::test_qwen3tts_06b_width_frame_loop_and_real_decoder" -m gpu -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 300 python tools/bench_qwen3tts.py 32 100 16 > gpurun_out/q3_norm.json 2>/dev/null; tail -1 gpurun_out/q3_norm.json | cut -c100-700
