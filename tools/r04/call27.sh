# round 4, GPU call 27: the XS switch test (bit identity against MIS_ATTN_XS=0) and Whisper-large-v3 at full depth on the landed build
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
( timeout 170 python -m pytest tests/test_gpu_whisper.py tests/test_gpu_fulldepth.py -k "cross_attention or whisper" -m gpu -x -q ) 2>&1 | grep -E "passed|failed|error|Error" | tail -4 | tee $O/c27_pytest.txt
