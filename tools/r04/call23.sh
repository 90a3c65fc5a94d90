# round 4, GPU call 23: the K/V stream of the decode attention non-temporal (nt) against default-policy loads, alternating
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
L=$(pwd)/mlx-audio-swift_amd/libmi_speech_nt.so
( MIS_LIB_PATH=$L timeout 300 python -m pytest tests/test_gpu_lm.py -k "orpheus or context or attn or teacher" -m gpu -x -q ) 2>&1 | grep -E "passed|failed|error" | tail -1
timeout 900 python tools/ab_decode.py $O/c23_ab.json n1:MIS_LIB_PATH=$L b2: n2:MIS_LIB_PATH=$L b3: n3:MIS_LIB_PATH=$L > $O/c23_ab.log 2>&1
python3 - <<'PY'
import json
rows = json.load(open("gpurun_out/r04/c23_ab.json"))
for r in rows: print(r["name"], r.get("step_ms"), r.get("value"), (r.get("kernels_us") or {}).get("attn_decode_ctx368"), r.get("error"))
b = [r["step_ms"] for r in rows if not r["env"] and "step_ms" in r]; p = [r["step_ms"] for r in rows if r["env"] and "step_ms" in r]
print("default policy mean", sum(b) / max(len(b), 1), "nt mean", sum(p) / max(len(p), 1))
PY
