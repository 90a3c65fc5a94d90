"""Pin the Whisper oracle against HF transformers' WhisperModel (independent implementation), float32."""
import numpy as np
import torch

from oracle import whisper as ow


def _hf(cfg, W):
    from transformers import WhisperConfig as HFC, WhisperForConditionalGeneration
    hc = HFC(vocab_size=cfg.vocab_size, num_mel_bins=cfg.num_mel_bins, d_model=cfg.d_model,
             encoder_layers=cfg.encoder_layers, encoder_attention_heads=cfg.encoder_attention_heads,
             encoder_ffn_dim=cfg.encoder_ffn_dim, max_source_positions=cfg.max_source_positions,
             decoder_layers=cfg.decoder_layers, decoder_attention_heads=cfg.decoder_attention_heads,
             decoder_ffn_dim=cfg.decoder_ffn_dim, max_target_positions=cfg.max_target_positions,
             scale_embedding=False, attn_implementation="eager", pad_token_id=0, bos_token_id=1, eos_token_id=2,
             decoder_start_token_id=3, suppress_tokens=None, begin_suppress_tokens=None)
    m = WhisperForConditionalGeneration(hc).to(torch.float32).eval()
    sd = {k: v.to(torch.float32) for k, v in W.items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not [k for k in missing if "k_proj.bias" not in k], missing
    return m


def test_encoder_and_decoder_match_hf_fp32():
    cfg = ow.TINY
    W = ow.make_synthetic_weights(cfg, dtype=torch.float32)
    hf = _hf(cfg, W)
    rng = np.random.default_rng(0)
    feats = (rng.standard_normal((2, 3000, cfg.num_mel_bins)) * 0.5).astype(np.float32)
    toks = rng.integers(0, cfg.vocab_size, (2, 7))
    with torch.no_grad():
        enc_ref = hf.model.encoder(torch.from_numpy(feats).transpose(1, 2)).last_hidden_state.numpy()
        logits_ref = hf(input_features=torch.from_numpy(feats).transpose(1, 2),
                        decoder_input_ids=torch.from_numpy(toks)).logits.numpy()
    o = ow.WhisperOracle(cfg, W, round=None)
    o.reset(2)
    enc = o.encode(feats)
    for b in range(2):
        np.testing.assert_allclose(enc[b].numpy(), enc_ref[b], rtol=2e-4, atol=2e-4)
    # prefill 4 tokens then 3 single steps through the caches
    got = [o.decode([toks[0, :4], toks[1, :4]])]
    for t in range(4, 7):
        got.append(o.decode([toks[0, t:t + 1], toks[1, t:t + 1]]))
    for b in range(2):
        full = np.concatenate([g[b].numpy() for g in got], 0)
        np.testing.assert_allclose(full, logits_ref[b], rtol=3e-4, atol=3e-4)
    assert np.abs(logits_ref).max() > 1.0


def test_suppress_masks_and_bf16_mode():
    l = np.zeros(10, np.float32)
    out = ow.apply_suppress(l, 0, [1], [2, 3], 8)
    assert (out < -1e8).nonzero()[0].tolist() == [1, 2, 3, 8, 9]
    assert (ow.apply_suppress(l, 1, [1], [2, 3], 8) < -1e8).nonzero()[0].tolist() == [2, 3, 8, 9]
    cfg = ow.TINY
    W = ow.make_synthetic_weights(cfg)
    a = ow.WhisperOracle(cfg, W, round="bf16"); a.reset(1)
    b = ow.WhisperOracle(cfg, W, round=None); b.reset(1)
    f = (np.random.default_rng(1).standard_normal((1, 3000, 80)) * 0.5).astype(np.float32)
    ea, eb = a.encode(f)[0], b.encode(f)[0]
    assert (ea - eb).abs().max().item() < 0.05 * eb.abs().max().item()
    la, lb = a.decode([[5, 6, 7]])[0], b.decode([[5, 6, 7]])[0]
    assert (la - lb).abs().max().item() < 0.05 * lb.abs().max().item()


def test_large_v3_width_matches_hf_fp32():
    """The same pin at Whisper-large-v3's per-layer dimensions (d 1280, 20 heads x 64, ffn 5120, 128 mel bins; 1 + 1 layers, vocabulary
    cut to 8 000 rows so that the CPU suite stays quick): encoder output and teacher-forced decoder logits against HF, float32."""
    cfg = ow.WhisperConfig(vocab_size=8000, num_mel_bins=128, d_model=1280, encoder_layers=1, encoder_attention_heads=20, encoder_ffn_dim=5120,
                           decoder_layers=1, decoder_attention_heads=20, decoder_ffn_dim=5120)
    W = ow.make_synthetic_weights(cfg, dtype=torch.float32)
    hf = _hf(cfg, W)
    rng = np.random.default_rng(1)
    feats = (rng.standard_normal((1, 3000, cfg.num_mel_bins)) * 0.5).astype(np.float32)
    toks = rng.integers(0, cfg.vocab_size, (1, 6))
    with torch.no_grad():
        enc_ref = hf.model.encoder(torch.from_numpy(feats).transpose(1, 2)).last_hidden_state.numpy()
        logits_ref = hf(input_features=torch.from_numpy(feats).transpose(1, 2), decoder_input_ids=torch.from_numpy(toks)).logits.numpy()
    o = ow.WhisperOracle(cfg, W, round=None)
    o.reset(1)
    enc = o.encode(feats)
    np.testing.assert_allclose(enc[0].numpy(), enc_ref[0], rtol=3e-4, atol=3e-4)
    got = [o.decode([toks[0, :3]])] + [o.decode([toks[0, t:t + 1]]) for t in range(3, 6)]
    full = np.concatenate([g[0].numpy() for g in got], 0)
    np.testing.assert_allclose(full, logits_ref[0], rtol=5e-4, atol=5e-4)
