# round 4, GPU call 3: GEMM lab split-factor sweep (rebuilt), sampler phase stamps, A/B of the new defaults, full-depth parity tests
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
LAB_SWEEP_S=1 timeout 120 tools/gemm_lab/gemm_lab 32 64 > $O/c3_gemm_lab_sweep_rows32.jsonl 2> $O/c3_gemm_lab.err
python3 - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r04/c3_gemm_lab_sweep_rows32.jsonl") if l.startswith("{")]
for sh in ("qkv", "o_proj", "down", "lm_head"):
    rs = sorted([r for r in rows if r["shape"] == sh], key=lambda r: r["us"])
    print(sh, [(r["variant"], r["us"], r["max_rel_vs_product"]) for r in rs[:8]], "product", [r["us"] for r in rs if r["variant"].startswith("product")])
PY
tail -2 $O/c3_gemm_lab.err
timeout 120 python tools/samp_phases.py 32 2> $O/c3_samp_phases.txt; grep SAMP_DBG $O/c3_samp_phases.txt
timeout 400 python tools/ab_decode.py $O/c3_ab.json graph8:MIS_GRAPH_STEPS=8 down_k4:MIS_ARR_DOWN=2,4,4,8 > $O/c3_ab.log 2>&1
cat $O/c3_ab.log
( time timeout 1200 python -m pytest tests/test_gpu_fulldepth.py -m gpu -q --durations=5 ) > $O/c3_pytest_fulldepth.txt 2>&1
tail -25 $O/c3_pytest_fulldepth.txt
grep -h "depth\|real_dims" gpurun_out/parity_observed.jsonl | tail -8
