"""Experiment: how fast do the decode GEMMs run when their weights are already in the 256 MB Infinity Cache (same layer every
launch) compared with streaming from HBM (rotating over the 28 layers)?  Decides whether a prefetch branch in the step graph pays."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import mlx_audio_swift_amd as mas
    cfg = mas.LlamaTTSConfiguration(rope_theta=500000.0, rope_scaling={"factor": 32.0, "rope_type": "llama3"})
    lm = mas.LlamaTTSModel.synthetic(cfg, seed=1)
    out = {}
    for i, n in enumerate(["qkv", "o_proj", "gate_up", "down"]):
        ms, by = lm.time_gemm(i, 32, iters=56)
        out[n] = {"us": round(ms * 1e3, 2), "GBps": round(by / ms / 1e6, 1)}
    print(json.dumps(out))
else:
    for mode, env in (("hbm_rotating", {}), ("mall_fixed_layer", {"MIS_TIME_GEMM_FIXED_LAYER": "3"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        print(mode, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
