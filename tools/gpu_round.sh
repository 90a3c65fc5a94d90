set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fullwidth.py -m gpu -q 2>&1 | grep -v "^PARITY" | tail -8
grep split gpurun_out/parity_observed.jsonl | tail -4
