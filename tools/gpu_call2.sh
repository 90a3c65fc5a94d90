set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c2
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_generate.py tests/test_gpu_sampler.py tests/test_gpu_soprano.py "tests/test_gpu_fullwidth.py::test_orpheus_3b_width_teacher_forced_b32_contexts_40_400_705" -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > gpurun_out/c2/pytest_lm.txt
cat gpurun_out/c2/pytest_lm.txt
MIS_PF=0,16384,16384,0,15360,18432 timeout 600 python -m pytest tests/test_gpu_lm.py tests/test_gpu_generate.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 > gpurun_out/c2/pytest_lm_pf.txt
cat gpurun_out/c2/pytest_lm_pf.txt
cp gpurun_out/parity_observed.jsonl gpurun_out/c2/ 2>/dev/null
timeout 1500 python tools/ab_decode.py gpurun_out/c2/ab.json \
  att_o:MIS_PF=0,0,0,0,0,18432 \
  g2_qkv15:MIS_PF=0,0,0,0,15360,0 \
  g1_down15:MIS_PF=0,0,15360,0,0,0 \
  att_g1_g2:MIS_PF=0,0,15360,0,15360,18432 \
  qkv_o:MIS_PF=18432,0,0,0,0,0 \
  o_g1_down:MIS_PF=0,16384,16384,0,0,0 \
  all:MIS_PF=0,16384,16384,0,15360,18432 \
  att_g8:MIS_PF=0,0,8192,0,8192,18432 \
  att_g15_b128:MIS_PF=0,0,15360,0,15360,18432,MIS_PF_BLOCKS=128 \
  att_g24:MIS_PF=0,0,24576,0,24576,18432 \
  down_qkv:MIS_PF=0,0,0,30720,0,0 2>&1 | tail -20
