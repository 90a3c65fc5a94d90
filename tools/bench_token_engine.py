"""Laboratory bench of the batch-1 token engine (csrc/token_engine.hip) at Soprano-80M's real dimensions (17 layers, vocabulary 8192):
24-token prompt + 64 greedy steps in ONE persistent launch on 1 and on 2 XCDs, next to the product's launch chain on the same handle
(teacher-forced decode steps through mis_lm_forward are not a fair clock - the product's figure is the generate call of
tools/bench_soprano.py / bench.py's secondary block).  Prints one JSON line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas

cfg = mas.SopranoConfiguration(stop_token_id=-1)
lm = mas.LlamaTTSModel.synthetic(cfg.lm_configuration(), seed=4321)
rng = np.random.default_rng(1235)
prompt = rng.integers(4, 8000, 24).astype(np.int32)
N_NEW = 64
out = {"workload": "Soprano-80M LM (17 layers, d 512, ffn 2304, vocab 8192), batch 1, 24 prompt positions + 64 greedy steps, one persistent launch"}
weights = 2.0 * (17 * (768 * 512 + 512 * 512 + 2 * 2304 * 512 + 512 * 2304) + 8192 * 512)
for xcds in (1, 2, 4, 8):
    best = 1e9
    toks = None
    for rep in range(4):
        r = lm.debug_token_engine(prompt, N_NEW, xcds=xcds)
        best = min(best, r["ms"]) if rep else best
        toks = r["next_tokens"]
    n = len(prompt) + N_NEW
    out[f"xcds_{xcds}"] = {"ms": best, "us_per_position": best * 1e3 / n, "weights_GBps": weights * n / (best * 1e-3) / 1e9,
                           "audio_s_per_s_lm_only": 64 * 2048 / 32000.0 / (best * 1e-3 * 64 / n), "first_generated": [int(t) for t in toks[23:31]]}
# the launch chain's greedy continuation on the same weights (token parity of the two engines on a full-depth model)
lm.lm_reset(1, 128)
seq = list(prompt)
t0 = time.perf_counter()
for t in range(len(prompt) + N_NEW - 1):
    lg = lm.lm_forward(np.asarray(seq[t:t + 1], np.int32))[0]
    if t >= len(prompt) - 1:
        seq.append(int(lg.argmax()))
engine_toks = [int(t) for t in toks[len(prompt) - 1:len(prompt) - 1 + N_NEW]]
out["launch_chain_greedy_equal_tokens"] = int(sum(int(a == b) for a, b in zip(seq[len(prompt):], engine_toks)))
out["launch_chain_first_generated"] = seq[len(prompt):len(prompt) + 8]
print(json.dumps(out))

# the generate form as the product runs it (prompt through the launch chain's batched prefill, sampler inside the launch): device time of
# the persistent launch against the wall time of the whole call (allocations, prefill, K/V import, read-back)
gp = mas.GenerateParameters(max_tokens=N_NEW, temperature=0.7, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30, seed=7, sampler_flavor=1)
best_wall, best_dev = 1e9, 1e9
for rep in range(4):
    t0 = time.perf_counter()
    r = lm.debug_token_engine(prompt, N_NEW, xcds=4, sampling=gp, stop_id=-1, want_hidden=True)
    dt = time.perf_counter() - t0
    if rep:
        best_wall, best_dev = min(best_wall, dt * 1e3), min(best_dev, r["ms"])
print(json.dumps({"generate_form_4xcd": {"launch_ms": best_dev, "call_wall_ms": best_wall, "positions_in_launch": r["positions"] - (len(prompt) - 1), "chosen": r["chosen"]}}))
