set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_encodec.py "tests/test_gpu_fullwidth.py::test_dac_24khz_and_encodec_24khz_real_dims" -m gpu -q 2>&1 | grep -v "^PARITY" | tail -30
