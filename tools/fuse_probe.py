"""Experiment: does the fused-producer GEMM make progress at the bench shape?  One decode step, error flag and timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = mas.LlamaTTSConfiguration(num_hidden_layers=L, rope_theta=500000.0, rope_scaling={"factor": 32.0, "rope_type": "llama3"})
lm = mas.LlamaTTSModel.synthetic(cfg, seed=1)
lm.lm_reset(B, 64)
ids = np.arange(B, dtype=np.int32)
for i in range(3):
    t0 = time.perf_counter()
    try:
        lm.lm_forward(ids)
        print("step", i, "ok", round((time.perf_counter() - t0) * 1e3, 2), "ms", flush=True)
    except Exception as e:
        print("step", i, "ERR", e, round((time.perf_counter() - t0) * 1e3, 2), "ms", flush=True)
if len(sys.argv) > 3:
    from mlx_audio_swift_amd.synthetic import snac_synthetic_weights
    sc = mas.SNACConfig()
    codec = mas.SNAC.from_weights(sc, snac_synthetic_weights(sc, seed=1234))
    lm2 = mas.LlamaTTSModel.synthetic(cfg, codec=codec, seed=1)
    rng = np.random.default_rng(0)
    prompts = [np.concatenate([[128259], rng.integers(0, 128000, 8), [128009, 128260, 128257]]).astype(np.int32) for _ in range(B)]
    gp = mas.GenerateParameters(max_tokens=int(sys.argv[3]), temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=1, frame_constrained=True)
    for i in range(2):
        t0 = time.perf_counter()
        try:
            pcm = lm2.generate_batch(prompts, gp)
            print("generate", i, "ok", round((time.perf_counter() - t0) * 1e3, 1), "ms", len(pcm[0]), flush=True)
        except Exception as e:
            print("generate", i, "ERR", e, round((time.perf_counter() - t0) * 1e3, 1), "ms", flush=True)
