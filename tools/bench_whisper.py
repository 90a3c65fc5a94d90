"""Secondary bench (BASELINE configs[3] shape, one GPU's share): Whisper-large-v3, synthetic weights, 8 x 30 s windows,
mel + encoder + 4-token prompt + 96 decode steps (EOT suppressed by using an out-of-range eot id)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = mas.WhisperConfig(vocab_size=51866, num_mel_bins=128, d_model=1280, encoder_layers=32, encoder_attention_heads=20,
                        encoder_ffn_dim=5120, decoder_layers=32, decoder_attention_heads=20, decoder_ffn_dim=5120)
m = mas.WhisperModel.synthetic(cfg, seed=777)
rng = np.random.default_rng(0)
wins = [(0.1 * rng.standard_normal(480000)).astype(np.float32) for _ in range(B)]
gp = mas.STTGenerateParameters(max_tokens=96, temperature=0.0, eot_id=-1, timestamp_begin=50365)
prompt = [50258, 50259, 50360, 50364]
feats = mas.dsp.whisper_encoder_features(np.stack(wins), 128)
for rep in range(2):
    t0 = time.perf_counter(); m.encode(feats, want_output=False); t_enc = time.perf_counter() - t0
    t0 = time.perf_counter(); ids = m.transcribe_windows(wins, prompt, gp); t_all = time.perf_counter() - t0
flops_enc = 2.27e12 * B
print(json.dumps({"workload": f"whisper-large-v3 bf16, {B} x 30 s, 96 decode steps", "encode_ms": t_enc * 1e3,
                  "encoder_TFLOPs": flops_enc / t_enc / 1e12, "transcribe_ms": t_all * 1e3,
                  "audio_s_per_s": 30.0 * B / t_all, "tokens": [len(i) for i in ids][:4]}))
