set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/final/pytest_gpu.txt; cat gpurun_out/final/pytest_gpu.txt
cp gpurun_out/parity_observed.jsonl gpurun_out/final/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/final/bench.log 2>&1; tail -1 gpurun_out/final/bench.log > gpurun_out/final/bench.json; python -c "import json; j=json.load(open('gpurun_out/final/bench.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic_source']['matches_current_build'], j['phases_ms'], j['cpu_baseline']['value'])"
timeout 300 python tools/bench_qwen3tts.py 32 100 16 > gpurun_out/final/q3_bf16.json 2>/dev/null; tail -1 gpurun_out/final/q3_bf16.json | cut -c1-600
timeout 300 python tools/bench_soprano.py 32 > gpurun_out/final/soprano_b32.json 2>/dev/null; tail -1 gpurun_out/final/soprano_b32.json
