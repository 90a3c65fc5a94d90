"""CPU: pins for oracle/encodec.py: torch.nn.LSTM (same gate order), the reference's padding index rule, overlap-add."""
import numpy as np
import torch

from oracle import encodec as oe


def test_lstm_matches_torch_lstm():
    cfg = oe.TINY
    W = oe.make_synthetic_weights(cfg)
    m = oe.EncodecOracle(cfg, W)
    dim = 2 ** len(cfg.upsampling_ratios) * cfg.num_filters
    x = torch.randn(2, 9, dim, generator=torch.Generator().manual_seed(0))
    ref = torch.nn.LSTM(dim, dim, batch_first=True)
    p = "decoder.layers.1.lstm.0"
    with torch.no_grad():
        ref.weight_ih_l0.copy_(torch.from_numpy(W[p + ".Wx"])); ref.weight_hh_l0.copy_(torch.from_numpy(W[p + ".Wh"]))
        ref.bias_ih_l0.copy_(torch.from_numpy(W[p + ".bias"])); ref.bias_hh_l0.zero_()
        want = ref(x)[0]
    assert (m.lstm(p, x) - want).abs().max() < 1e-5


def test_reflect_padding_rule_and_causal_lengths():
    x = torch.arange(10, dtype=torch.float32).reshape(1, 1, 10)
    assert oe.pad1d(x, 3, 2, "reflect")[0, 0].tolist() == [3, 2, 1] + list(range(10)) + [8, 7]
    assert oe.pad1d(x[..., :2], 3, 0, "reflect")[0, 0].tolist() == [1, 1, 1, 0, 1]          # clamp for short inputs (:143-147)
    assert torch.equal(oe.pad1d(x, 3, 0, "reflect"), torch.nn.functional.pad(x, (3, 0), mode="reflect"))
    cfg = oe.TINY
    m = oe.EncodecOracle(cfg, oe.make_synthetic_weights(cfg))
    codes = np.random.default_rng(0).integers(0, cfg.codebook_size, (2, 2, 7))
    wav = m.decode_frame(codes)
    assert wav.shape == (2, 7 * cfg.hop_length) and np.isfinite(wav).all() and wav.std() > 1e-4
    assert cfg.num_quantizers == 3 and oe.EncodecConfig().num_quantizers == 32 and oe.EncodecConfig().hop_length == 320


def test_linear_overlap_add_is_a_weighted_average():
    rng = np.random.default_rng(1)
    frames = [rng.standard_normal((2, 10)).astype(np.float32) for _ in range(3)]
    out = oe.linear_overlap_add(frames, 6)
    assert out.shape == (2, 22)
    assert np.allclose(out[:, :6], frames[0][:, :6], atol=1e-6)                 # only one frame covers these samples
    const = oe.linear_overlap_add([np.ones((1, 10), np.float32)] * 3, 6)
    assert np.allclose(const, 1.0, atol=1e-6)


def test_decoder_and_quantizer_match_hf_transformers():
    """Independent implementation: HF transformers' EncodecModel (the architecture the reference mirrors layer by layer,
    Encodec.swift:94-170) with the oracle's synthetic weights; float32, causal, weight_norm folded to g = ||v||."""
    from transformers import EncodecConfig as HFC, EncodecModel
    cfg = oe.TINY
    hc = HFC(audio_channels=1, num_filters=cfg.num_filters, kernel_size=cfg.kernel_size, num_residual_layers=cfg.num_residual_layers,
             dilation_growth_rate=cfg.dilation_growth_rate, codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
             hidden_size=cfg.hidden_size, num_lstm_layers=cfg.num_lstm_layers, residual_kernel_size=cfg.residual_kernel_size,
             use_causal_conv=True, pad_mode="reflect", last_kernel_size=cfg.last_kernel_size, trim_right_ratio=1.0, compress=cfg.compress,
             upsampling_ratios=list(cfg.upsampling_ratios), target_bandwidths=list(cfg.target_bandwidths),
             sampling_rate=cfg.sampling_rate, use_conv_shortcut=True, norm_type="weight_norm")
    hf = EncodecModel(hc).eval()
    W = oe.make_synthetic_weights(cfg, seed=7)
    sd = hf.state_dict()

    def put_conv(hf_prefix, o_prefix, transposed):
        w = torch.from_numpy(W[o_prefix + ".conv.weight"])
        w = w.permute(2, 0, 1).contiguous() if transposed else w.permute(0, 2, 1).contiguous()     # -> torch layouts
        g = sd[hf_prefix + ".conv.parametrizations.weight.original0"]
        norm_dims = tuple(range(1, w.ndim))
        sd[hf_prefix + ".conv.parametrizations.weight.original0"] = w.norm(dim=norm_dims, keepdim=True).reshape(g.shape)
        sd[hf_prefix + ".conv.parametrizations.weight.original1"] = w
        sd[hf_prefix + ".conv.bias"] = torch.from_numpy(W[o_prefix + ".conv.bias"])

    for name in [k for k in sd if k.startswith("decoder.") and k.endswith(".conv.bias")]:
        hp = name[: -len(".conv.bias")]
        layer = hf.get_submodule(hp)
        put_conv(hp, hp, type(layer).__name__ == "EncodecConvTranspose1d")
    for j in range(cfg.num_lstm_layers):
        p = f"decoder.layers.1.lstm.{j}"
        sd[f"decoder.layers.1.lstm.weight_ih_l{j}"] = torch.from_numpy(W[p + ".Wx"])
        sd[f"decoder.layers.1.lstm.weight_hh_l{j}"] = torch.from_numpy(W[p + ".Wh"])
        sd[f"decoder.layers.1.lstm.bias_ih_l{j}"] = torch.from_numpy(W[p + ".bias"])
        sd[f"decoder.layers.1.lstm.bias_hh_l{j}"] = torch.zeros_like(sd[f"decoder.layers.1.lstm.bias_hh_l{j}"])
    nq = cfg.num_quantizers
    for i in range(nq):
        sd[f"quantizer.layers.{i}.codebook.embed"] = torch.from_numpy(W[f"quantizer.layers.{i}.codebook.embed"])
    hf.load_state_dict(sd)
    o = oe.EncodecOracle(cfg, W)
    rng = np.random.default_rng(3)
    codes = rng.integers(0, cfg.codebook_size, (2, nq, 11))
    with torch.no_grad():
        z_hf = hf.quantizer.decode(torch.from_numpy(codes).transpose(0, 1))          # HF: [nq, B, T]
        y_hf = hf.decoder(z_hf)
    z = o.quantizer_decode(codes)
    assert torch.allclose(z, z_hf, atol=1e-6)
    y = o.decode_frame(codes)
    assert y.shape == (2, 11 * cfg.hop_length)
    np.testing.assert_allclose(y, y_hf[:, 0].numpy(), rtol=1e-4, atol=2e-5)


def test_group_norm_stereo_variant_matches_hf_transformers():
    """The 48 kHz model family (norm_type time_group_norm, non-causal padding, two audio channels) against HF EncodecModel."""
    from transformers import EncodecConfig as HFC, EncodecModel
    cfg = oe.TINY_48K
    hc = HFC(audio_channels=2, num_filters=cfg.num_filters, kernel_size=cfg.kernel_size, num_residual_layers=cfg.num_residual_layers,
             dilation_growth_rate=cfg.dilation_growth_rate, codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
             hidden_size=cfg.hidden_size, num_lstm_layers=cfg.num_lstm_layers, residual_kernel_size=cfg.residual_kernel_size,
             use_causal_conv=False, pad_mode="reflect", last_kernel_size=cfg.last_kernel_size, trim_right_ratio=1.0, compress=cfg.compress,
             upsampling_ratios=list(cfg.upsampling_ratios), target_bandwidths=list(cfg.target_bandwidths),
             sampling_rate=cfg.sampling_rate, use_conv_shortcut=True, norm_type="time_group_norm", normalize=True,
             chunk_length_s=1.0, overlap=0.01)
    hf = EncodecModel(hc).eval()
    W = oe.make_synthetic_weights(cfg, seed=8)
    sd = hf.state_dict()
    for name in [k for k in sd if k.startswith("decoder.") and k.endswith(".conv.bias")]:
        hp = name[: -len(".conv.bias")]
        transposed = type(hf.get_submodule(hp)).__name__ == "EncodecConvTranspose1d"
        w = torch.from_numpy(W[hp + ".conv.weight"])
        sd[hp + ".conv.weight"] = w.permute(2, 0, 1).contiguous() if transposed else w.permute(0, 2, 1).contiguous()
        sd[hp + ".conv.bias"] = torch.from_numpy(W[hp + ".conv.bias"])
        sd[hp + ".norm.weight"] = torch.from_numpy(W[hp + ".norm.weight"])
        sd[hp + ".norm.bias"] = torch.from_numpy(W[hp + ".norm.bias"])
    for j in range(cfg.num_lstm_layers):
        p = f"decoder.layers.1.lstm.{j}"
        sd[f"decoder.layers.1.lstm.weight_ih_l{j}"] = torch.from_numpy(W[p + ".Wx"])
        sd[f"decoder.layers.1.lstm.weight_hh_l{j}"] = torch.from_numpy(W[p + ".Wh"])
        sd[f"decoder.layers.1.lstm.bias_ih_l{j}"] = torch.from_numpy(W[p + ".bias"])
        sd[f"decoder.layers.1.lstm.bias_hh_l{j}"] = torch.zeros_like(sd[f"decoder.layers.1.lstm.bias_hh_l{j}"])
    nq = cfg.num_quantizers
    for i in range(nq):
        sd[f"quantizer.layers.{i}.codebook.embed"] = torch.from_numpy(W[f"quantizer.layers.{i}.codebook.embed"])
    hf.load_state_dict(sd)
    o = oe.EncodecOracle(cfg, W)
    codes = np.random.default_rng(4).integers(0, cfg.codebook_size, (2, nq, 13))
    with torch.no_grad():
        y_hf = hf.decoder(hf.quantizer.decode(torch.from_numpy(codes).transpose(0, 1)))
    y = o.decode_frame(codes)
    assert y.shape == (2, 2, 13 * cfg.hop_length)
    np.testing.assert_allclose(y, y_hf.numpy(), rtol=1e-4, atol=2e-5)
