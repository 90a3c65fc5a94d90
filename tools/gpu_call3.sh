set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c3
export TMPDIR=/tmp
timeout 1500 python tools/ab_decode.py gpurun_out/c3/ab.json \
  att_o:MIS_PF=0,0,0,0,0,18432 \
  g2_qkv15:MIS_PF=0,0,0,0,15360,0 \
  g1_down15:MIS_PF=0,0,15360,0,0,0 \
  att_g1_g2:MIS_PF=0,0,15360,0,15360,18432 \
  qkv_o:MIS_PF=18432,0,0,0,0,0 \
  o_g1_down:MIS_PF=0,16384,16384,0,0,0 \
  all:MIS_PF=0,16384,16384,0,15360,18432 \
  att_g8:MIS_PF=0,0,8192,0,8192,18432 \
  att_g15_b128:MIS_PF=0,0,15360,0,15360,18432+MIS_PF_BLOCKS=128 \
  att_g24:MIS_PF=0,0,24576,0,24576,18432 \
  down_qkv:MIS_PF=0,0,0,30720,0,0 2>&1 | tail -20
