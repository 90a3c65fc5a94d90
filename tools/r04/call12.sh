# round 4, GPU call 12: MIS_ATTN_PAIR settled with alternating runs (the first A/B was inside the run-to-run noise)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python tools/ab_decode.py $O/c12_ab.json p1:MIS_ATTN_PAIR=1 b2: p2:MIS_ATTN_PAIR=1 b3: p3:MIS_ATTN_PAIR=1 > $O/c12_ab.log 2>&1
python3 - <<'PY'
import json
rows = json.load(open("gpurun_out/r04/c12_ab.json"))
for r in rows: print(r["name"], r.get("step_ms"), r.get("value"))
b = [r["step_ms"] for r in rows if not r["env"]]; p = [r["step_ms"] for r in rows if r["env"]]
print("base mean", sum(b) / len(b), "pair mean", sum(p) / len(p))
PY
