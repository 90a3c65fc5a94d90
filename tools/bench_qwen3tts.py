"""Secondary bench (BASELINE configs[4] / SURVEY §8d C5, one GPU's share): Qwen3-TTS-0.6B-shaped synthetic model (bf16 weights; the
8-bit checkpoint format dequantises to this), batch 32, 100 frames (8 s) per row, EOS out of reach, chunked audio delivery every
25 frames.  argv[1] = batch (default 32), argv[2] = frames (default 100)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas
from mlx_audio_swift_amd.synthetic import qwen3tts_synthetic_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = mas.Qwen3TTSConfiguration(codec_eos_token_id=3071)            # inside the suppressed range but exempt: never the argmax in practice
t0 = time.perf_counter()
m = mas.Qwen3TTSModel(cfg)
for name, arr in qwen3tts_synthetic_weights(cfg):
    m.set_tensor(name, arr)
m.finalize()
t_load = time.perf_counter() - t0
rng = np.random.default_rng(1238)
prompts = []
for b in range(B):
    text = rng.integers(0, 151000, 24)
    t = list(text[:3]) + [cfg.tts_pad_token_id] * 3 + [cfg.tts_bos_token_id] + [int(text[3])]
    c = [-1, -1, -1, cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id, cfg.codec_pad_id, cfg.codec_bos_id]
    prompts.append(mas.PreparedPrompt(np.asarray(t, np.int32), np.asarray(c, np.int32),
                                      np.asarray(list(text[4:]) + [cfg.tts_eos_token_id], np.int32), 0))
gp = mas.Qwen3TTSGenerateParameters(max_tokens=F, temperature=0.9, top_k=50, repetition_penalty=1.05, seed=9)
res = {}
for rep in range(2):
    t0 = time.perf_counter(); codes = m.generate_codes(prompts, gp); t_codes = time.perf_counter() - t0
    t0 = time.perf_counter(); pcm = m.generate_batch(prompts, gp); t_all = time.perf_counter() - t0
audio_s = sum(len(p) for p in pcm) / cfg.sample_rate
print(json.dumps({"workload": f"Qwen3-TTS-0.6B-shaped bf16, batch {B}, {F} frames/row, 16 code groups, speech-tokenizer decode per row",
                  "load_s": t_load, "frames": [len(c) for c in codes][:4], "codes_ms": t_codes * 1e3,
                  "ms_per_frame": t_codes * 1e3 / F, "generate_ms": t_all * 1e3, "decode_ms": (t_all - t_codes) * 1e3,
                  "audio_s_per_s": audio_s / t_all, "audio_s": audio_s}))
