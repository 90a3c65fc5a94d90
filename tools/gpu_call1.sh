set -x
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.jsonl
timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_generate.py tests/test_gpu_sampler.py "tests/test_gpu_fullwidth.py::test_orpheus_3b_width_teacher_forced_b32_contexts_40_400_705" -m gpu -x -q 2>&1 | tail -5 > gpurun_out/c1/pytest_lm.txt
cat gpurun_out/c1/pytest_lm.txt
timeout 900 python -m pytest tests/test_gpu_depth.py -m gpu -q 2>&1 | tail -15 > gpurun_out/c1/pytest_depth.txt
cat gpurun_out/c1/pytest_depth.txt
cp gpurun_out/parity_observed.jsonl gpurun_out/c1/ 2>/dev/null
timeout 1500 python tools/ab_decode.py gpurun_out/c1/ab.json \
  glue_old:MIS_GLUE_V4=0 \
  pf_fork_only:MIS_PREFETCH=7,MIS_PREFETCH_CAP_KB=64 \
  pf1:MIS_PREFETCH=1 pf2:MIS_PREFETCH=2 pf4:MIS_PREFETCH=4 pf8:MIS_PREFETCH=8 pf16:MIS_PREFETCH=16 \
  pf5:MIS_PREFETCH=5 pf7:MIS_PREFETCH=7 pf13:MIS_PREFETCH=13 pf15:MIS_PREFETCH=15 \
  pf7_b64:MIS_PREFETCH=7,MIS_PREFETCH_BLOCKS=64 pf7_b1024:MIS_PREFETCH=7,MIS_PREFETCH_BLOCKS=1024 2>&1 | tail -20
