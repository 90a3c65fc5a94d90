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
