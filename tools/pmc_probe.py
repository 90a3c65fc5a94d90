"""Workload for the rocprofv3 --pmc passes over the decode-step kernels (tools/pmc_traffic.sh): every kernel of the Orpheus-3B
step chain launched 28 times over rotating layers (so the 256 MB Infinity Cache cannot serve the operands) at batch 32 -
gate+up (the dominant kernel, 100.7 MB algorithmic bytes), lm_head, the split-K GEMMs (qkv, o_proj, down), decode attention at
context 368 and the slab-reduce / residual / RMSNorm kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas
cfg = mas.LlamaTTSConfiguration(rope_theta=500000.0, rope_scaling={"factor": 32.0, "rope_type": "llama3"})
lm = mas.LlamaTTSModel.synthetic(cfg, seed=1)
lm.lm_reset(32, 512)
names = ["qkv", "o_proj", "gate_up", "down", "lm_head", "attn_decode_ctx368", "reduce_residual_rmsnorm"]
for which in (2, 4, 0, 1, 3, 5, 6):
    ms, by = lm.time_gemm(which, 32, 28)
    print(names[which], round(ms * 1e3, 2), "us", round(by / ms / 1e6, 1), "GB/s", by)
