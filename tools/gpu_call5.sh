set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c5
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
timeout 600 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_generate.py -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|assert|FAILED" | tail -12 > gpurun_out/c5/pytest_sampler.txt
cat gpurun_out/c5/pytest_sampler.txt
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_sampler.py --deselect tests/test_gpu_generate.py 2>&1 | grep -E "passed|failed|error|Error|FAILED" | tail -12 > gpurun_out/c5/pytest_rest.txt
cat gpurun_out/c5/pytest_rest.txt
cp gpurun_out/parity_observed.jsonl gpurun_out/c5/ 2>/dev/null
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/c5/bench.log 2>&1; tail -1 gpurun_out/c5/bench.log | cut -c1-1500
