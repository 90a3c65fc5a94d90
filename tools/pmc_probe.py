"""Launch the dominant GEMM (gate+up, 100.7 MB algorithmic bytes) 28 times over rotating layers for a
rocprofv3 --pmc pass (FETCH_SIZE / WRITE_SIZE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas
cfg = mas.LlamaTTSConfiguration(rope_theta=500000.0, rope_scaling={"factor": 32.0, "rope_type": "llama3"})
lm = mas.LlamaTTSModel.synthetic(cfg, seed=1)
lm.lm_reset(32, 64)
for which in (2, 4):
    ms, by = lm.time_gemm(which, 32, 28)
    print(which, round(by / ms / 1e6, 1), "GB/s", by)
