"""Prompt latency of long prompts at small batch (Orpheus voice-cloning prompts carry the reference recording's SNAC codes: 7 tokens per
12 Hz frame, ~840 tokens for 10 s): mis_lm_prefill on the Orpheus-3B shape, batched [positions x rows] pass against position by position.
argv[1] = prompt tokens (default 840), argv[2] = batch (default 1), argv[3] = layers (default 28)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas

P = int(sys.argv[1]) if len(sys.argv) > 1 else 840
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = int(sys.argv[3]) if len(sys.argv) > 3 else 28
cfg = mas.LlamaTTSConfiguration(num_hidden_layers=L, rope_theta=500000.0, rope_scaling={"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                                                                      "original_max_position_embeddings": 8192, "rope_type": "llama3"})
lm = mas.LlamaTTSModel.synthetic(cfg, seed=4321)
rng = np.random.default_rng(0)
rows = [rng.integers(0, 128000, P - (b % 5)).astype(np.int32) for b in range(B)]
out = {"workload": f"Orpheus-3B shape ({L} layers), batch {B}, {P}-token prompts, mis_lm_prefill"}
for mode, name in (("0", "batched_ms"), ("1", "position_by_position_ms")):
    os.environ["MIS_PREFILL_SEQ"] = mode
    if mode == "1" and P * L > 12000:
        continue                                  # ~2 ms per position and layer group: skip the long sequential run
    lm.lm_prefill(rows, max_context=P + 8)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); lm.lm_prefill(rows, max_context=P + 8); ts.append(time.perf_counter() - t0)
    out[name] = round(min(ts) * 1e3, 2)
print(json.dumps(out))
