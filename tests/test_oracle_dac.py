"""CPU: pins for oracle/dac.py: the reference's own shape tests and a naive transposed-conv restatement."""
import numpy as np
import torch

from oracle import dac as od


def test_reference_shape_pins():
    # Tests/MLXAudioCodecsTests.swift:1127-1194
    assert od.num_samples(od.DacConfig(decoder_rates=(8, 5, 4, 2)), 250) == 80_043
    assert od.num_samples(od.DacConfig(decoder_rates=(8, 5, 4, 2)), 375) == 120_043
    assert od.num_samples(od.DacConfig(decoder_rates=(8, 8, 4, 2)), 430) == 220_235


def test_transposed_conv_with_output_padding_matches_naive():
    rng = np.random.default_rng(0)
    cfg = od.TINY
    W = od.make_synthetic_weights(cfg)
    m = od.DacOracle(cfg, W)
    p = "decoder.model.1.block.1"
    s, cin, cout, T = 3, 48, 24, 5
    x = rng.standard_normal((1, cin, T)).astype(np.float32)
    got = m.convt(p, torch.from_numpy(x), s).numpy()[0]
    v, g, b = W[p + ".weight_v"].astype(np.float64), W[p + ".weight_g"].astype(np.float64), W[p + ".bias"]
    w = g * v / (np.sqrt((v * v).sum(axis=(0, 1), keepdims=True)) + 1e-12)            # norm over all axes except the input one
    full = np.zeros((cout, (T - 1) * s + 2 * s))
    for n in range(T):
        for j in range(2 * s):
            full[:, n * s + j] += w[:, j, :] @ x[0, :, n]
    pad = 2                                                                             # ceil(3 / 2)
    ref = full[:, pad: full.shape[1] - (pad - 1)] + b[:, None]
    assert got.shape == ref.shape == (cout, od.convt_out_len(T, s)) and np.abs(got - ref).max() < 1e-5


def test_decode_shapes_and_range():
    cfg = od.TINY
    m = od.DacOracle(cfg, od.make_synthetic_weights(cfg))
    codes = np.random.default_rng(1).integers(0, cfg.codebook_size, (2, cfg.n_codebooks, 6))
    wav = m.decode_from_codes(codes)
    assert wav.shape == (2, od.num_samples(cfg, 6)) and np.abs(wav).max() <= 1.0 and wav.std() > 1e-3
