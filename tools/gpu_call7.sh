set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c7
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_qwen3tts.py tests/test_gpu_soprano.py tests/test_gpu_loader.py tests/test_gpu_generate.py "tests/test_gpu_fullwidth.py::test_qwen3tts_06b_width_frame_loop_and_real_decoder" -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|assert|FAILED" | tail -14 > gpurun_out/c7/pytest.txt
cat gpurun_out/c7/pytest.txt
cp gpurun_out/parity_observed.jsonl gpurun_out/c7/ 2>/dev/null
timeout 300 python tools/bench_qwen3tts.py 32 100 16 > gpurun_out/c7/q3_fused.json 2>/dev/null; cat gpurun_out/c7/q3_fused.json
MIS_FUSE_NORM=0 timeout 300 python tools/bench_qwen3tts.py 32 100 16 > gpurun_out/c7/q3_unfused.json 2>/dev/null; cat gpurun_out/c7/q3_unfused.json
timeout 300 python tools/bench_soprano.py 32 > gpurun_out/c7/sop_fused.json 2>/dev/null; cat gpurun_out/c7/sop_fused.json
MIS_FUSE_NORM=0 timeout 300 python tools/bench_soprano.py 32 > gpurun_out/c7/sop_unfused.json 2>/dev/null; cat gpurun_out/c7/sop_unfused.json
timeout 300 python tools/bench_soprano.py 1 > gpurun_out/c7/sop1_fused.json 2>/dev/null; cat gpurun_out/c7/sop1_fused.json
