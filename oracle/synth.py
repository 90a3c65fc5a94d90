"""Deterministic synthetic tensors shared by oracle, tests, bench and the HIP library.

Test infrastructure (the HIP library has its own device-side generator implementing the
same documented formula, csrc/lm_engine.hip: mis_synth_fill).  There is no network, hence no
checkpoints: every weight in tests and in bench.py is synthetic.

Formula ("mis-synth-v1"): element i of tensor with 64-bit key k is
    u = splitmix64(k * 0x9E3779B97F4A7C15 + i)            (one round, below)
    x = ((u >> 40) + 0.5) * 2^-24   in (0,1)                (24 random bits)
    value = (2x - 1) * amplitude                            (uniform in (-amp, amp))
computed in float32 and (for bf16 tensors) rounded to nearest-even bf16.
"""
from __future__ import annotations

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(key: int, n: int, start: int = 0) -> np.ndarray:
    """24-bit uniforms in (0,1) as float32, element index start..start+n."""
    with np.errstate(over="ignore"):
        base = np.uint64(key) * np.uint64(0x9E3779B97F4A7C15)
        idx = np.arange(start, start + n, dtype=np.uint64) + base
    u = splitmix64(idx)
    return ((u >> np.uint64(40)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def _uniform01_torch(key: int, n: int, start: int = 0):
    """uniform01 on torch int64 tensors (multi-threaded; two's-complement wrap-around = uint64 arithmetic, logical right shifts
    emulated by masking).  Bit-identical to the numpy form (tests/test_oracle_llama.py)."""
    import torch

    def s64(v):                                   # python int (mod 2^64) -> signed 64-bit value
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    base = s64(key * 0x9E3779B97F4A7C15)
    out = torch.empty(n, dtype=torch.float32)
    CH = 1 << 18                                  # cache-resident chunks: 20x faster than one pass over 200 MB temporaries
    for c0 in range(0, n, CH):
        m = min(CH, n - c0)
        z = torch.arange(start + c0, start + c0 + m, dtype=torch.int64) + base
        z = z + s64(0x9E3779B97F4A7C15)
        z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
        z = z ^ lsr(z, 31)
        out[c0:c0 + m] = (lsr(z, 40).to(torch.float32) + 0.5) * float(2.0 ** -24)
    return out


def synth_tensor(key: int, shape, amplitude: float) -> np.ndarray:
    n = int(np.prod(shape))
    if n >= (1 << 20):                            # large tensors: torch path (same bits, many cores)
        try:
            x = _uniform01_torch(key, n)
            return ((2.0 * x - 1.0) * float(np.float32(amplitude))).numpy().reshape(shape)
        except ImportError:                       # pragma: no cover
            pass
    x = uniform01(key, n)
    return ((np.float32(2.0) * x - np.float32(1.0)) * np.float32(amplitude)).reshape(shape)


def synth_rows(key: int, shape, amplitude: float, rows) -> np.ndarray:
    """Selected rows of the 2-D tensor synth_tensor(key, shape, amplitude) without materialising it (the generator is
    addressable by element index): full-vocabulary tests need a few thousand of the 156 940 embedding rows."""
    ncol = int(shape[1])
    out = np.empty((len(rows), ncol), np.float32)
    for j, r in enumerate(rows):
        x = uniform01(key, ncol, start=int(r) * ncol)
        out[j] = (np.float32(2.0) * x - np.float32(1.0)) * np.float32(amplitude)
    return out


def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest-even bfloat16 -> float32 (numpy, no torch)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    b = x.view(np.uint32)
    rounding = ((b >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    with np.errstate(over="ignore"):
        r = (b + rounding) & np.uint32(0xFFFF0000)
    return r.view(np.float32)
