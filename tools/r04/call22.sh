# round 4, GPU call 22: split-K slabs stored write-through (sc1) against plain stores, alternating (does the kernel boundary pay for dirty slabs?)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
L=$(pwd)/mlx-audio-swift_amd/libmi_speech_sc1.so
timeout 900 python tools/ab_decode.py $O/c22_ab.json w1:MIS_LIB_PATH=$L b2: w2:MIS_LIB_PATH=$L b3: w3:MIS_LIB_PATH=$L > $O/c22_ab.log 2>&1
python3 - <<'PY'
import json
rows = json.load(open("gpurun_out/r04/c22_ab.json"))
for r in rows: print(r["name"], r.get("step_ms"), r.get("value"), r.get("kernels_us") or r.get("error"))
b = [r["step_ms"] for r in rows if not r["env"] and "step_ms" in r]; p = [r["step_ms"] for r in rows if r["env"] and "step_ms" in r]
print("plain stores mean", sum(b) / max(len(b), 1), "write-through mean", sum(p) / max(len(p), 1))
PY
