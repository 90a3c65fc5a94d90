# round 4, GPU call 13: glue with three column groups per thread (256-thread blocks at d = 3072): parity + alternating A/B
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_generate.py -k "orpheus or full_size or shards" -m gpu -x -q ) > $O/c13_pytest.txt 2>&1; grep -E "passed|failed|error" $O/c13_pytest.txt | tail -2
timeout 900 python tools/ab_decode.py $O/c13_ab.json c1:MIS_GLUE_CPT=1 b2: c2:MIS_GLUE_CPT=1 b3: c3:MIS_GLUE_CPT=1 > $O/c13_ab.log 2>&1
python3 - <<'PY'
import json
rows = json.load(open("gpurun_out/r04/c13_ab.json"))
for r in rows: print(r["name"], r.get("step_ms"), r.get("value"), r.get("kernels_us", {}).get("reduce_residual_rmsnorm"))
b = [r["step_ms"] for r in rows if not r["env"]]; p = [r["step_ms"] for r in rows if r["env"]]
print("three groups per thread (default) mean", sum(b) / len(b), "one group per thread mean", sum(p) / len(p))
PY
