"""CPU pins for two numerical building blocks of the codec kernels (no GPU, no library call):
 - mis_sin_sq (csrc/codec_kernels.h): the 12-instruction sin^2 of the Snake activations, restated operation by operation in float32
 - the split-bf16 contraction of csrc/codec_bf3.hip (x = xh + xl, three products, f32 accumulation) inside the SNAC oracle."""
import numpy as np

from oracle import snac as osn

F = np.float32


def sin_sq_f32(y):
    """mis_sin_sq, same constants and operation order; numpy float32 arithmetic (fma emulated in float64 and rounded once)."""
    y = np.asarray(y, F)

    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)
    n = np.rint(y * F(0.318309886183790672))
    r = fma(n, np.full_like(y, F(-3.140625)), y)
    r = fma(n, np.full_like(y, F(-9.67653589793e-4)), r)
    s = r * r
    p = fma(s, np.full_like(y, F(-2.50521084e-8)), np.full_like(y, F(2.75573192e-6)))
    p = fma(s, p, np.full_like(y, F(-1.98412698e-4)))
    p = fma(s, p, np.full_like(y, F(8.33333333e-3)))
    p = fma(s, p, np.full_like(y, F(-1.66666667e-1)))
    p = fma(r * s, p, r)
    return p * p


def test_sin_sq_polynomial_accuracy():
    y = (np.arange(-2_000_000, 2_000_001, dtype=np.float64) * 1.37e-4).astype(F)          # |y| <= 274, far beyond Snake's alpha * x
    ref = np.sin(y.astype(np.float64)) ** 2
    err = np.abs(sin_sq_f32(y).astype(np.float64) - ref)
    assert err.max() < 3.5e-7                                                              # f32 rounding of the polynomial and of p * p; flat in |y|


def _bf16_round(a):
    u = np.ascontiguousarray(a, F).view(np.uint32)
    return ((u + (((u >> 16) & 1) + 0x7FFF)) & 0xFFFF0000).view(F)


def test_split_bf16_contraction_keeps_the_waveform_gate():
    """Three bf16 products per f32 product inside every dense conv of the SNAC oracle: the waveform moves by ~3e-5 of its RMS
    (the 24 kHz configuration, tools/bf16x3_emulation.py: 1.06e-5 absolute against the 1e-4 gate); plain bf16 operands do not."""
    cfg = osn.SnacConfig(**osn.TINY)
    o = osn.SnacOracle(cfg, osn.make_synthetic_weights(cfg))
    codes, noise = osn.synthetic_codes(cfg, 1, 6), osn.synthetic_noise(cfg, 1, 6)
    ref = o.decode(codes, noise)
    einsum = np.einsum

    def run(terms):
        def split_einsum(spec, a, b):
            ah, bh = _bf16_round(a), _bf16_round(b)
            if terms == 1:
                return einsum(spec, ah, bh).astype(F)
            al, bl = _bf16_round(a - ah), _bf16_round(b - bh)
            return (einsum(spec, ah, bh) + einsum(spec, ah, bl) + einsum(spec, al, bh)).astype(F)
        np.einsum = split_einsum
        try:
            return o.decode(codes, noise)
        finally:
            np.einsum = einsum
    rms = lambda d: float(np.sqrt(np.mean(np.asarray(d, np.float64) ** 2)))
    e3, e1, scale = rms(run(3) - ref), rms(run(1) - ref), rms(ref)
    assert e3 < 1e-4 * scale and e3 < 3e-5 and e1 > 30 * e3, (e3, e1, scale)
