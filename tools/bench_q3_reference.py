"""Reference-audio front end of Qwen3-TTS in-context voice cloning at the published widths (ECAPA-TDNN 512 / 1536 channels on 128 mels,
Mimi encoder 64 filters / 8 x 512 transformer / 16 kept codebooks of 2048 x 256), synthetic weights: time per recording of S seconds.
argv[1] = seconds (default 10)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas
from mlx_audio_swift_amd.synthetic import qwen3tts_reference_synthetic_weights, qwen3tts_synthetic_weights

S = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
cfg = mas.Qwen3TTSConfiguration(codec_eos_token_id=3071)
cfg.speaker_encoder = mas.qwen3tts.Qwen3TTSSpeakerEncoderConfiguration()
cfg.tokenizer_encoder = mas.qwen3tts.Qwen3TTSTokenizerEncoderConfiguration()
m = mas.Qwen3TTSModel(cfg)
for name, arr in qwen3tts_synthetic_weights(cfg):
    m.set_tensor(name, arr)
for name, arr in qwen3tts_reference_synthetic_weights(cfg):
    m.set_tensor(name, arr)
m.finalize()
n = int(24000 * S)
rng = np.random.default_rng(0)
t = np.arange(n) / 24000.0
audio = (0.25 * np.sin(2 * np.pi * 180 * t) + 0.05 * rng.standard_normal(n)).astype(np.float32)
res = {"workload": f"{S:g} s reference recording at 24 kHz, published front-end widths, one MI355X"}
for name, fn in (("speaker_embedding_ms", m.extract_speaker_embedding), ("encode_audio_ms", m.encode_audio)):
    fn(audio)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); out = fn(audio); ts.append(time.perf_counter() - t0)
    res[name] = round(min(ts) * 1e3, 2)
res["frames"] = int(out.shape[1])
t0 = time.perf_counter(); ctx = m.add_reference(out, m.extract_speaker_embedding(audio)); res["add_reference_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
print(json.dumps(res))
