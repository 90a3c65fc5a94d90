"""Phase stamps of the one-launch full-vocabulary sampler (k_samp_cluster): block (0, 0)'s s_memtime at every phase boundary and the
time of 20 back-to-back launches, printed by the library on stderr when MIS_SAMP_DBG is set (csrc/lm_engine.hip, mis_sample_logits).
Usage: python tools/samp_phases.py [rows=32]   (stamps are counter ticks relative to kernel entry; 100 ticks = 1 us on gfx950)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MIS_SAMP_DBG"] = "1"
import mlx_audio_swift_amd as mas
from mlx_audio_swift_amd.tts import sample_logits

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
V, ctx = 156940, 20
rng = np.random.default_rng(0)
logits = (rng.standard_normal((B, V)) * 2.0).astype(np.float32)
window = rng.integers(0, V, (B, ctx)).astype(np.int32)
wl = np.full(B, ctx, np.int32)
for name, kw in (("nucleus", dict(temperature=0.6, top_p=0.8)), ("no_nucleus", dict(temperature=1.0, top_p=1.0)), ("greedy", dict(temperature=0.0, top_p=0.8))):
    p = mas.GenerateParameters(repetition_penalty=1.3, seed=1, row_offset=0, frame_constrained=2, **kw)
    sys.stderr.write(name + " ")
    sys.stderr.flush()
    sample_logits(logits, window, wl, p, 3)
