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
# frame_constrained = 2: the bench's stand-in (every id visited, ids outside the step's frame slot get zero mass - their exp and histogram
# updates cost less than live ids'); frame_constrained = 0: every id live, what an unconstrained checkpoint pays
for name, kw in (("nucleus", dict(temperature=0.6, top_p=0.8, frame_constrained=2)), ("no_nucleus", dict(temperature=1.0, top_p=1.0, frame_constrained=2)),
                 ("greedy", dict(temperature=0.0, top_p=0.8, frame_constrained=2)), ("nucleus_all_ids_live", dict(temperature=0.6, top_p=0.8, frame_constrained=0)),
                 ("nucleus_all_ids_live_peaked", dict(temperature=0.6, top_p=0.8, frame_constrained=0, peaked=True))):
    lg = logits
    if kw.pop("peaked", False):                     # a trained model's logits: a few ids carry the mass (the level-1 histogram then has few hot bins)
        lg = logits.copy(); lg[:, 1000:1008] += 14.0
    p = mas.GenerateParameters(repetition_penalty=1.3, seed=1, row_offset=0, **kw)
    sys.stderr.write(name + " ")
    sys.stderr.flush()
    sample_logits(lg, window, wl, p, 3)
