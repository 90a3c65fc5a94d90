"""Qwen3-TTS prompt latency: an in-context (voice-cloning) prompt is a few hundred positions - [reference text + target text] over
codec_pad, then the reference recording's frames (12.5 per second).  Times generate_codes with two frames after a P-position prompt, the
batched prefill (lm_prefill.hip on the talker's packed weights) against the position-by-position graph (MIS_PREFILL_SEQ=1), 0.6B shape."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas
from mlx_audio_swift_amd.synthetic import qwen3tts_synthetic_weights

cfg = mas.Qwen3TTSConfiguration(codec_eos_token_id=3071)
BITS = int(sys.argv[1]) if len(sys.argv) > 1 else 16          # 16 = bf16 weights, 8 / 4 = every 2-D talker tensor MLX affine-quantised
m = mas.Qwen3TTSModel(cfg)
for name, arr in qwen3tts_synthetic_weights(cfg):
    if BITS != 16 and arr.ndim == 2 and not name.startswith("decoder.") and arr.shape[1] % 64 == 0:
        from mlx_audio_swift_amd.synthetic import mlx_affine_quantize
        wq, sc, bi = mlx_affine_quantize(arr, 64, BITS)
        m.set_quantized_tensor(name, wq, sc, bi, 64, BITS)
    else:
        m.set_tensor(name, arr)
m.finalize()
rng = np.random.default_rng(3)
gp = mas.Qwen3TTSGenerateParameters(max_tokens=2, temperature=0.0)
out = {}
for B, P in ((1, 350), (1, 40), (32, 40), (8, 350)):
    prompts = []
    for b in range(B):
        n = P - (b % 3)
        t = rng.integers(0, 151000, n).astype(np.int32)
        c = np.full(n, cfg.codec_pad_id, np.int32)
        c[:3] = -1
        prompts.append(mas.PreparedPrompt(t, c, np.zeros(0, np.int32), 0))
    res = {}
    for mode in ("0", "1"):
        os.environ["MIS_PREFILL_SEQ"] = mode
        m.generate_codes(prompts, gp)
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); codes = m.generate_codes(prompts, gp); ts.append(time.perf_counter() - t0)
        res["sequential_ms" if mode == "1" else "batched_ms"] = round(min(ts) * 1e3, 2)
        res["codes_" + mode] = [int(x) for x in codes[0][0][:4]]
    out[f"batch{B}_prompt{P}"] = res
out["weights_bits"] = BITS
print(json.dumps(out))
