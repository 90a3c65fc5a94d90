cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06; mkdir -p $O
echo "== V1=1 bench"; MIS_ATTN_PREFILL_V1=1 timeout 120 python tools/bench_whisper.py 2>&1 | grep -v amdgpu.ids | head -30 | cut -c1-300
echo "== fold test B=3, old encoder kernel"; MIS_ATTN_PREFILL_V1=1 timeout 200 python -m pytest tests/test_gpu_whisper.py -m gpu -q -x -k "glue_folded and 3" 2>&1 | grep -v amdgpu.ids | head -40 | cut -c1-300
echo "== bit identity test"; timeout 200 python -m pytest tests/test_gpu_whisper.py -m gpu -q -x -k "256_row" 2>&1 | grep -v amdgpu.ids | head -40 | cut -c1-300
