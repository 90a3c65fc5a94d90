# round 4, GPU call 2: split-factor sweep in the GEMM lab, the rewritten one-launch sampler (parity + failure path), arrangement A/B
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
LAB_SWEEP_S=1 timeout 120 tools/gemm_lab/gemm_lab 32 64 > $O/c2_gemm_lab_sweep_rows32.jsonl 2> $O/c2_gemm_lab.err
python3 - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r04/c2_gemm_lab_sweep_rows32.jsonl") if l.startswith("{")]
for sh in ("qkv", "o_proj", "down", "lm_head"):
    rs = sorted([r for r in rows if r["shape"] == sh], key=lambda r: r["us"])
    print(sh, [(r["variant"], r["us"], r["max_rel_vs_product"]) for r in rs[:6]], "product", [r["us"] for r in rs if r["variant"].startswith("product")])
PY
( time timeout 900 python -m pytest tests/test_gpu_sampler.py tests/test_gpu_generate.py tests/test_gpu_lm.py tests/test_gpu_whisper.py -m gpu -x -q --durations=5 ) > $O/c2_pytest.txt 2>&1
tail -12 $O/c2_pytest.txt
timeout 600 python tools/ab_decode.py $O/c2_ab.json graph1:MIS_GRAPH_STEPS=1 arr_r03:MIS_ARR_TUNED=0 down_r2k2:MIS_ARR_DOWN=2,2,4,8 o_r2k2:MIS_ARR_O=2,2,4,2 qkv_r4k2:MIS_ARR_QKV=4,2,3,3 six_kernel_sampler:MIS_SAMPLER_WIDE=1 > $O/c2_ab.log 2>&1
cat $O/c2_ab.log
