# round 4, GPU call 15: glue at d = 3072 with six groups per thread (128 threads) against three (256 threads), alternating
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
( MIS_GLUE_CPT=6 timeout 600 python -m pytest tests/test_gpu_fullwidth.py -k "orpheus" -m gpu -x -q ) 2>&1 | grep -E "passed|failed|error" | tail -1
timeout 900 python tools/ab_decode.py $O/c15_ab.json s1:MIS_GLUE_CPT=6 b2: s2:MIS_GLUE_CPT=6 b3: s3:MIS_GLUE_CPT=6 > $O/c15_ab.log 2>&1
python3 - <<'PY'
import json
rows = json.load(open("gpurun_out/r04/c15_ab.json"))
for r in rows: print(r["name"], r.get("step_ms"), r.get("value"))
b = [r["step_ms"] for r in rows if not r["env"]]; p = [r["step_ms"] for r in rows if r["env"]]
print("three groups (default) mean", sum(b) / len(b), "six groups mean", sum(p) / len(p))
PY
