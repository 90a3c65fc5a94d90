set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_observed.jsonl
timeout 1500 python -m pytest tests -m gpu -q -rP 2>&1 | grep -v "^PARITY\|^---\|^$" | tail -60 > gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 300 python tools/mall_probe.py > gpurun_out/mall_probe.txt 2>&1; cat gpurun_out/mall_probe.txt
