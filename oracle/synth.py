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


def synth_tensor(key: int, shape, amplitude: float) -> np.ndarray:
    n = int(np.prod(shape))
    x = uniform01(key, n)
    return ((np.float32(2.0) * x - np.float32(1.0)) * np.float32(amplitude)).reshape(shape)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest-even bfloat16 -> float32 (numpy, no torch)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    b = x.view(np.uint32)
    rounding = ((b >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    with np.errstate(over="ignore"):
        r = (b + rounding) & np.uint32(0xFFFF0000)
    return r.view(np.float32)
