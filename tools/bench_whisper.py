"""Secondary bench (BASELINE configs[3] shape, one GPU's share): Whisper-large-v3, synthetic weights, 8 x 30 s windows,
mel + encoder + 4-token prompt + 96 decode steps (EOT suppressed by using an out-of-range eot id).
argv[1] = windows per GPU (default 8); --gpus N: N replicas (one per visible GPU, same synthetic weights), N x windows sharded inside
the library by mis_whisper_group_generate (configs[3] as worded: 64 windows over 8 GPUs)."""
import json, os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas

NG = 1
if "--gpus" in sys.argv:
    k = sys.argv.index("--gpus")
    NG = int(sys.argv[k + 1])
    del sys.argv[k:k + 2]
B = (int(sys.argv[1]) if len(sys.argv) > 1 else 8) * NG
cfg = mas.WhisperConfig(vocab_size=51866, num_mel_bins=128, d_model=1280, encoder_layers=32, encoder_attention_heads=20,
                        encoder_ffn_dim=5120, decoder_layers=32, decoder_attention_heads=20, decoder_ffn_dim=5120)
reps = [mas.WhisperModel.synthetic(cfg, device=i, seed=777) for i in range(NG)]
m = reps[0]
rng = np.random.default_rng(0)
wins = [(0.1 * rng.standard_normal(480000)).astype(np.float32) for _ in range(B)]
gp = mas.STTGenerateParameters(max_tokens=96, temperature=0.0, eot_id=-1, timestamp_begin=50365)
prompt = [50258, 50259, 50360, 50364]
feats = mas.dsp.whisper_encoder_features(np.stack(wins[: B // NG]), 128)
for rep in range(2):
    t0 = time.perf_counter(); m.encode(feats, want_output=False); t_enc = time.perf_counter() - t0
    t0 = time.perf_counter(); ids = m.transcribe_windows(wins, prompt, gp, replicas=reps if NG > 1 else None); t_all = time.perf_counter() - t0
flops_enc = 2.27e12 * (B // NG)
print(json.dumps({"workload": f"whisper-large-v3 bf16, {B} x 30 s over {NG} GPU(s), 96 decode steps", "n_gpus": NG, "encode_ms": t_enc * 1e3,
                  "encoder_TFLOPs": flops_enc / t_enc / 1e12, "transcribe_ms": t_all * 1e3,
                  "audio_s_per_s": 30.0 * B / t_all, "tokens": [len(i) for i in ids][:4],
                  "token_crc32": zlib.crc32(np.concatenate([np.asarray(i, np.int64) for i in ids]).tobytes()),       # A/B runs: same tokens?
                  "MIS_ATTN_XS": os.environ.get("MIS_ATTN_XS")}))
