set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c6
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
timeout 900 python -m pytest tests/test_gpu_generate.py tests/test_gpu_qwen3tts.py tests/test_gpu_whisper.py tests/test_gpu_soprano.py -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|assert|FAILED" | tail -14 > gpurun_out/c6/pytest.txt
cat gpurun_out/c6/pytest.txt
cp gpurun_out/parity_observed.jsonl gpurun_out/c6/ 2>/dev/null
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/c6/bench.log 2>&1; tail -1 gpurun_out/c6/bench.log | cut -c1-2500
timeout 300 python tools/bench_qwen3tts.py 32 100 16 > gpurun_out/c6/q3_bf16.json 2>/dev/null; cat gpurun_out/c6/q3_bf16.json
