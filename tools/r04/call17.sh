# round 4, GPU call 17: o_proj as R2 KSB2 (128-thread blocks, same split) against R2 KSB4, alternating
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python tools/ab_decode.py $O/c17_ab.json k1:MIS_ARR_O=2,2,4,2 b2: k2:MIS_ARR_O=2,2,4,2 b3: k3:MIS_ARR_O=2,2,4,2 > $O/c17_ab.log 2>&1
python3 - <<'PY'
import json
rows = json.load(open("gpurun_out/r04/c17_ab.json"))
for r in rows: print(r["name"], r.get("step_ms"), r.get("value"), r.get("kernels_us", {}).get("o_proj"))
b = [r["step_ms"] for r in rows if not r["env"]]; p = [r["step_ms"] for r in rows if r["env"]]
print("KSB4 (default) mean", sum(b) / len(b), "KSB2 mean", sum(p) / len(p))
PY
