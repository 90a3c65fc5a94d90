set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c9
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_qwen3tts.py tests/test_gpu_soprano.py tests/test_gpu_loader.py "tests/test_gpu_fullwidth.py::test_qwen3tts_06b_width_frame_loop_and_real_decoder" -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|assert|FAILED" | tail -14 > gpurun_out/c9/pytest.txt
cat gpurun_out/c9/pytest.txt
cp gpurun_out/parity_observed.jsonl gpurun_out/c9/ 2>/dev/null
timeout 300 python tools/bench_qwen3tts.py 32 100 16 > gpurun_out/c9/q3_fused.json 2>/dev/null; cat gpurun_out/c9/q3_fused.json
timeout 300 python tools/bench_qwen3tts.py 32 100 8 > gpurun_out/c9/q3_fused_8bit.json 2>/dev/null; cat gpurun_out/c9/q3_fused_8bit.json
MIS_FUSE_NORM=0 timeout 300 python tools/bench_qwen3tts.py 32 100 8 > gpurun_out/c9/q3_unfused_8bit.json 2>/dev/null; cat gpurun_out/c9/q3_unfused_8bit.json
timeout 300 python tools/bench_soprano.py 32 > gpurun_out/c9/sop_fused.json 2>/dev/null; cat gpurun_out/c9/sop_fused.json
R=$PWD
rm -rf /tmp/ks1; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -- python $R/tools/bench_qwen3tts.py 32 40 16 > /tmp/ks1.log 2>&1)
cp $(find /tmp/ks1 -name "*kernel_stats.csv" | head -1) gpurun_out/c9/q3_kernel_stats_fused_wide.csv
head -8 gpurun_out/c9/q3_kernel_stats_fused_wide.csv | cut -c1-160
