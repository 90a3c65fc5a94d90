# round 4, GPU call 25: (a) split-K slab reads non-temporal (glue + attention prologue; alternate build) against the landed build, alternating;
# (b) the decode attention against the context length (what the tile imbalance between waves costs) - diagnostics for the next round
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
L=$(pwd)/mlx-audio-swift_amd/libmi_speech_v1.so
timeout 600 python tools/ab_decode.py $O/c25_ab.json v1:MIS_LIB_PATH=$L b2: v2:MIS_LIB_PATH=$L > $O/c25_ab.log 2>&1
python3 - <<'PY'
import json
rows = json.load(open("gpurun_out/r04/c25_ab.json"))
for r in rows: print(r["name"], r.get("step_ms"), r.get("value"), (r.get("kernels_us") or {}).get("reduce_residual_rmsnorm"), (r.get("kernels_us") or {}).get("attn_decode_ctx368"), r.get("error"))
PY
timeout 300 python tools/attn_ctx_sweep.py $O/c25_attn_ctx.json --layers 16 224 256 288 368 512 544 2>&1 | tail -8
