"""Sweep the work decomposition of the weight-streaming GEMMs at the bench shape (Orpheus-3B, B=32):
in-block split (KSB) and inter-block split-K (S).  Prints GB/s per setting; writes gpurun_out/gemm_sweep.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas  # noqa: E402

cfg = mas.LlamaTTSConfiguration(rope_theta=500000.0, rope_scaling={"factor": 32.0, "rope_type": "llama3"})
lm = mas.LlamaTTSModel.synthetic(cfg, seed=1)
names = ["qkv", "o_proj", "gate_up", "down", "lm_head"]
res = []


def run(env, which):
    for k in list(os.environ):
        if k.startswith("MIS_"):
            del os.environ[k]
    os.environ.update({k: str(v) for k, v in env.items()})
    lm.lm_reset(31, 64); lm.lm_reset(32, 64)            # re-read the env
    out = {}
    for w in which:
        ms, by = lm.time_gemm(w, 32, iters=56)
        out[names[w]] = round(by / ms / 1e6, 1)
    res.append({"env": env, "GBps": out})
    print(env, out, flush=True)


for R in (1, 2):
    for ksb in (1, 4):
        for S in (1, 2, 3, 4, 6, 8, 12):
            run({"MIS_R_PART": R, "MIS_KSB_PART": ksb, "MIS_S_QKV": S, "MIS_S_O": S, "MIS_S_DOWN": S}, [0, 1, 3])
for ksb in (1, 4):
    run({"MIS_KSB_GU": ksb}, [2])
    run({"MIS_KSB_HEAD": ksb}, [4])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_sweep.json"), "w"), indent=1)
