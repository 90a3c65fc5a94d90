import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas
cfg = mas.LlamaTTSConfiguration(rope_theta=500000.0, rope_scaling={"factor": 32.0, "rope_type": "llama3"})
lm = mas.LlamaTTSModel.synthetic(cfg, seed=1)
names = ["qkv", "o_proj", "gate_up", "down", "lm_head"]
lm.lm_reset(32, 64)
for rep in range(2):
    print({n: round((lambda r: r[1] / r[0] / 1e6)(lm.time_gemm(i, 32, 56)), 1) for i, n in enumerate(names)}, flush=True)
