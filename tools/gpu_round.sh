set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/final/pytest_gpu.txt; cat gpurun_out/final/pytest_gpu.txt
cp gpurun_out/parity_observed.jsonl gpurun_out/final/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
