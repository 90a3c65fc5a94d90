"""Pin the Soprano decoder oracle against independent torch implementations of its pieces."""
import numpy as np
import torch
import torch.nn.functional as Fn

from oracle import soprano as osop


def test_interpolate_matches_torch_align_corners():
    x = np.random.default_rng(0).standard_normal((2, 5, 9)).astype(np.float32)
    got = osop.interpolate1d(x, 4 * 8 + 1)
    ref = Fn.interpolate(torch.from_numpy(x), size=33, mode="linear", align_corners=True).numpy()
    np.testing.assert_allclose(got, ref, atol=2e-6)
    assert osop.interpolate1d(x[:, :, :1], 7).shape == (2, 5, 7)


def test_backbone_matches_torch_and_istft_identities():
    cfg = osop.TINY
    W = osop.make_synthetic_weights(cfg)
    o = osop.SopranoDecoderOracle(cfg, W)
    x = np.random.default_rng(1).standard_normal((2, 13, cfg.hidden_size)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.asarray(a, np.float64))
    p = "decoder.decoder"
    h = Fn.conv1d(t(x).transpose(1, 2), t(W[p + ".embed.weight"]).permute(0, 2, 1), t(W[p + ".embed.bias"]),
                  padding=cfg.input_kernel // 2).transpose(1, 2)
    h = Fn.layer_norm(h, (cfg.decoder_dim,), t(W[p + ".norm.weight"]), t(W[p + ".norm.bias"]), 1e-6)
    for i in range(cfg.decoder_num_layers):
        q = f"{p}.convnext.{i}"
        r = h
        u = Fn.conv1d(h.transpose(1, 2), t(W[q + ".dwconv.weight"]).permute(0, 2, 1), t(W[q + ".dwconv.bias"]),
                      padding=cfg.dw_kernel // 2, groups=cfg.decoder_dim).transpose(1, 2)
        u = Fn.layer_norm(u, (cfg.decoder_dim,), t(W[q + ".norm.weight"]), t(W[q + ".norm.bias"]), 1e-6)
        u = Fn.gelu(u @ t(W[q + ".pwconv1.weight"]).T + t(W[q + ".pwconv1.bias"]))
        u = u @ t(W[q + ".pwconv2.weight"]).T + t(W[q + ".pwconv2.bias"])
        h = r + t(W[q + ".gamma"]) * u
    h = Fn.layer_norm(h, (cfg.decoder_dim,), t(W[p + ".final_layer_norm.weight"]), t(W[p + ".final_layer_norm.bias"]), 1e-6)
    np.testing.assert_allclose(o.backbone(x), h.numpy(), rtol=2e-4, atol=2e-4)
    # full decode: length (L-1)*upscale*hop, finite, non-degenerate
    y = o.decode(x)
    L = cfg.upscale * (13 - 1) + 1
    assert y.shape == (2, (L - 1) * cfg.hop_length) and np.isfinite(y).all() and y.std() > 1e-3
    # ISTFT: a pure DC spectrum of magnitude m gives frames of constant m/n_fft... normalised by the window SUM:
    # windowed-OLA of constant c with window w divided by sum(w) returns c wherever the window sum is non-zero
    cfg1 = osop.SopranoDecoderConfig(hidden_size=4, decoder_num_layers=0, decoder_dim=4, n_fft=16, hop_length=4)
    Wh = {"decoder.head.out.weight": np.zeros((18, 4), np.float32), "decoder.head.out.bias": np.zeros(18, np.float32)}
    Wh["decoder.head.out.bias"][0] = np.log(8.0)                      # mag[0] = 8 -> irfft constant 8/16 = 0.5
    Wh["decoder.head.out.bias"][1:9] = -30.0                          # other magnitudes ~ 0
    oh = osop.SopranoDecoderOracle(cfg1, Wh)
    a = oh.head(np.zeros((1, 6, 4), np.float32))
    assert a.shape == (1, 5 * 4) and np.allclose(a, 0.5, atol=1e-5)


def test_soprano_penalty_per_occurrence():
    l = np.float32([2.0, -2.0, 0.0, 4.0])
    out = osop.soprano_repetition_penalty(l, [0, 0, 1, 3], 2.0)
    assert out.tolist() == [0.5, -4.0, 0.0, 2.0]
