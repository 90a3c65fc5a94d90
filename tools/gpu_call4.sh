set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c4
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
timeout 600 python -m pytest tests/test_gpu_lm.py -m gpu -x -q -k "second_attention or teacher_forced or context_beyond or untied" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 > gpurun_out/c4/pytest_attn2.txt
cat gpurun_out/c4/pytest_attn2.txt
cp gpurun_out/parity_observed.jsonl gpurun_out/c4/ 2>/dev/null
timeout 1800 python tools/ab_decode.py gpurun_out/c4/ab.json \
  attn_v1:MIS_ATTN_V2=0 \
  att_o:MIS_PF=0,0,0,0,0,18432 \
  g2_qkv15:MIS_PF=0,0,0,0,15360,0 \
  g1_down15:MIS_PF=0,0,15360,0,0,0 \
  att_g1_g2:MIS_PF=0,0,15360,0,15360,18432 \
  qkv_o:MIS_PF=18432,0,0,0,0,0 \
  o_g1_down:MIS_PF=0,16384,16384,0,0,0 \
  all:MIS_PF=0,16384,16384,0,15360,18432 \
  att_g8:MIS_PF=0,0,8192,0,8192,18432 \
  att_g24:MIS_PF=0,0,24576,0,24576,18432 \
  v1_att_g1_g2:MIS_ATTN_V2=0+MIS_PF=0,0,15360,0,15360,18432 2>&1 | tail -20
