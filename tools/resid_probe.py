"""Per-launch time of the four block projections at the bench shape (Orpheus-3B, B=32) with the residual stream kept in the GEMMs
(MIS_RESID_IN_GEMM=1: norm applied to the operand in registers / residual epilogue with last-arriver reduce) against the classic
arrangement (plain operand / split-K slabs reduced by a separate launch).  Prints microseconds per launch."""
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


def run(env, which=(0, 1, 2, 3)):
    for k in list(os.environ):
        if k.startswith("MIS_"):
            del os.environ[k]
    os.environ.update({k: str(v) for k, v in env.items()})
    lm.lm_reset(31, 64); lm.lm_reset(32, 64)            # re-read the env
    out = {}
    for w in which:
        ms, by = lm.time_gemm(w, 32, iters=56)
        out[names[w]] = round(ms * 1e3, 2)
    res.append({"env": env, "us": out})
    print(env, out, flush=True)


run({"MIS_RESID_IN_GEMM": 0})
run({"MIS_RESID_IN_GEMM": 1})
for R in (1, 2):
    for ksb in (4, 8, 16):
        for S in (1, 2):
            run({"MIS_RESID_IN_GEMM": 1, "MIS_R_PART": R, "MIS_KSB_PART": ksb, "MIS_S_O": S, "MIS_S_DOWN": S}, (1, 3))
            run({"MIS_RESID_IN_GEMM": 0, "MIS_R_PART": R, "MIS_KSB_PART": ksb, "MIS_S_O": S, "MIS_S_DOWN": S}, (1, 3))
for S in (3, 4, 6):
    run({"MIS_RESID_IN_GEMM": 1, "MIS_S_O": S, "MIS_S_DOWN": S}, (1, 3))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "resid_probe.json"), "w"), indent=1)
