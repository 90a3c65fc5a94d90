"""Secondary bench (BASELINE configs[1] / SURVEY §8d C2): Soprano-80M-shaped synthetic model, fixed 24-token prompt,
64 forced decode steps ([STOP] out of range) -> 129 024 samples (4.03 s @ 32 kHz) per row.  argv[1] = batch (default 1)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = mas.SopranoConfiguration(stop_token_id=-1)
m = mas.SopranoModel.synthetic(cfg, seed=4321)
rng = np.random.default_rng(1235)
rows = [rng.integers(4, 8000, 24).astype(np.int32) for _ in range(B)]
gp = mas.GenerateParameters(max_tokens=64, temperature=0.7, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30,
                            seed=7, sampler_flavor=1)
best = 1e9
for rep in range(4):
    t0 = time.perf_counter(); pcm = m.generate_batch(rows, gp); dt = time.perf_counter() - t0
    best = min(best, dt)
audio_s = sum(len(p) for p in pcm) / cfg.sample_rate
print(json.dumps({"workload": f"Soprano-80M bf16 LM + f32 Vocos/ISTFT decoder, batch {B}, 24-token prompt, 64 new tokens",
                  "samples_per_row": int(len(pcm[0])), "generate_ms": best * 1e3, "audio_s_per_s": audio_s / best,
                  "ms_per_token": best * 1e3 / 64}))
