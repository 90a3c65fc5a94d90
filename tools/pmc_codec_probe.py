"""Workload for a rocprofv3 --pmc pass over the MFMA-bound kernels: Qwen3-TTS speech-tokenizer decode (k_conv_taps, k_snac_gemm),
SNAC decode (k_snac_gemm), Whisper-large-v3 encoder (k_gemm_big, k_attn_prefill), log-mel (k_mel_tile).  Synthetic weights."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas
from mlx_audio_swift_amd.synthetic import qwen3tts_synthetic_weights, snac_synthetic_weights

rng = np.random.default_rng(0)
what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("all", "q3"):
    cfg = mas.Qwen3TTSConfiguration(talker=mas.qwen3tts._lm(256, 1, 256, 2, 1, 128, 3072), predictor=mas.qwen3tts._lm(256, 1, 256, 2, 1, 128, 2048),
                                    text_hidden_size=128, text_vocab_size=1000, tts_pad_token_id=991)
    m = mas.Qwen3TTSModel(cfg)
    for name, arr in qwen3tts_synthetic_weights(cfg):
        m.set_tensor(name, arr)
    m.finalize()
    codes = rng.integers(0, 2048, (8, 16, 100)).astype(np.int32)
    for _ in range(2):
        m.decode_codes(codes)
if what in ("all", "snac"):
    sc = mas.SNACConfig()
    codec = mas.SNAC.from_weights(sc, snac_synthetic_weights(sc, seed=1234))
    codes = [rng.integers(0, 4096, (8, 96 * s)).astype(np.int32) for s in (1, 2, 4)]
    for _ in range(2):
        codec.decode(codes)
if what == "snac32":                                  # the bench's codec call: 32 rows x 96 groups
    sc = mas.SNACConfig()
    codec = mas.SNAC.from_weights(sc, snac_synthetic_weights(sc, seed=1234))
    codes = [rng.integers(0, 4096, (32, 96 * s)).astype(np.int32) for s in (1, 2, 4)]
    for _ in range(2):
        codec.decode(codes)
if what == "q3b32":                                   # C5's decode: 32 rows x 100 frames
    cfg = mas.Qwen3TTSConfiguration(talker=mas.qwen3tts._lm(256, 1, 256, 2, 1, 128, 3072), predictor=mas.qwen3tts._lm(256, 1, 256, 2, 1, 128, 2048),
                                    text_hidden_size=128, text_vocab_size=1000, tts_pad_token_id=991)
    m = mas.Qwen3TTSModel(cfg)
    for name, arr in qwen3tts_synthetic_weights(cfg):
        m.set_tensor(name, arr)
    m.finalize()
    codes = rng.integers(0, 2048, (32, 16, 100)).astype(np.int32)
    for _ in range(2):
        m.decode_codes(codes)
if what in ("all", "whisper"):
    wc = mas.WhisperConfig(vocab_size=51866, num_mel_bins=128, d_model=1280, encoder_layers=32, encoder_attention_heads=20,
                           encoder_ffn_dim=5120, decoder_layers=2, decoder_attention_heads=20, decoder_ffn_dim=5120)
    w = mas.WhisperModel.synthetic(wc, seed=777)
    wins = np.stack([(0.1 * rng.standard_normal(480000)).astype(np.float32) for _ in range(4)])
    feats = mas.dsp.whisper_encoder_features(wins, 128)
    for _ in range(2):
        w.encode(feats, want_output=False)
print("done")
