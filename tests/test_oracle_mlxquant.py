"""CPU: the affine quantisation restatement round-trips within a step and packs little-endian within the word."""
import numpy as np

from oracle import mlxquant as mq


def test_roundtrip_and_packing():
    rng = np.random.default_rng(0)
    w = rng.standard_normal((8, 128)).astype(np.float32)
    for bits in (2, 4, 8):
        wq, s, b = mq.quantize(w, 64, bits)
        assert wq.shape == (8, 128 * bits // 32) and s.shape == b.shape == (8, 2)
        d = mq.dequantize(wq, s, b, 64, bits)
        step = np.repeat(np.abs(s), 64, axis=1)
        assert np.all(np.abs(d - w) <= 1.01 * step + 1e-6)     # half a step inside the range, up to one at the clipped far edge
        # the extreme of larger magnitude in each group is reproduced exactly (mlx's edge refinement)
        g = w.reshape(8, 2, 64); dg = d.reshape(8, 2, 64)
        ext = np.where(np.abs(g.min(-1)) > np.abs(g.max(-1)), g.min(-1), g.max(-1))
        dext = np.take_along_axis(dg, np.abs(g).argmax(-1)[..., None], -1)[..., 0]
        assert np.allclose(dext, ext, rtol=1e-6)
    wq, s, b = mq.quantize(w, 64, 4)
    q0 = int(np.clip(np.round((w[0, 0] - b[0, 0]) / s[0, 0]), 0, 15)); q1 = int(np.clip(np.round((w[0, 1] - b[0, 0]) / s[0, 0]), 0, 15))
    assert (int(wq[0, 0]) & 0xF) == q0 and ((int(wq[0, 0]) >> 4) & 0xF) == q1


def test_quantized_matmul_is_the_matmul_with_the_float32_dequantised_weight():
    """quantized_matmul never rounds the weight: it equals x @ dequantize(...)^T up to float32 summation order - NOT the
    matmul with a bf16-rounded dequantised weight (what a dequantise-at-load engine computes)."""
    import torch
    rng = np.random.default_rng(3)
    w = rng.standard_normal((48, 256)).astype(np.float32) * 0.05
    x = torch.from_numpy(rng.standard_normal((5, 256)).astype(np.float32)).bfloat16().float().numpy()
    for bits in (4, 8):
        wq, s, b = mq.quantize(w, 64, bits)
        s = torch.from_numpy(s).bfloat16().float().numpy(); b = torch.from_numpy(b).bfloat16().float().numpy()
        y = mq.quantized_matmul(x, wq, s, b, 64, bits)
        d32 = mq.dequantize(wq, s, b, 64, bits)
        ref = x.astype(np.float64) @ d32.astype(np.float64).T
        assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-6
        d16 = torch.from_numpy(d32).bfloat16().float().numpy()
        worse = np.abs(x.astype(np.float64) @ d16.astype(np.float64).T - ref).max()
        assert worse > 20 * np.abs(y - ref).max()                          # the rounding the native kernel removes
