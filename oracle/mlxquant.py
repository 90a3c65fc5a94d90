"""MLX affine quantisation (mlx.core.quantize / dequantize [3P: mlx 0.31, not vendored]) restated for tests.
Test infrastructure only.  Layout: a [N, K] matrix becomes `wq` uint32 [N, K*bits/32] (element i of a row lives in word
i // (32/bits) at bit offset bits * (i % (32/bits))), `scales` and `biases` [N, K/group_size]; w ~= scales * q + biases.
The reference consumes this format through QuantizedLinear / QuantizedEmbedding (LlamaTTS.swift:958-968)."""
from __future__ import annotations

import numpy as np


def quantize(w: np.ndarray, group_size: int = 64, bits: int = 4):
    """mlx quantize: per group, scale = max((max - min) / (2^bits - 1), 1e-7) with the sign / edge refinement that makes the
    group's extreme value of larger magnitude exactly representable."""
    w = np.asarray(w, np.float32)
    N, K = w.shape
    assert K % group_size == 0 and 32 % bits == 0
    n_bins = float(2 ** bits - 1)
    g = w.reshape(N, K // group_size, group_size)
    w_max, w_min = g.max(-1), g.min(-1)
    mask = np.abs(w_min) > np.abs(w_max)
    scales = np.maximum((w_max - w_min) / n_bins, 1e-7).astype(np.float32)
    scales = np.where(mask, scales, -scales)
    edge = np.where(mask, w_min, w_max)
    q0 = np.round(edge / scales)
    scales = np.where(q0 != 0, edge / np.where(q0 != 0, q0, 1), scales).astype(np.float32)
    biases = np.where(q0 == 0, 0.0, edge).astype(np.float32)
    q = np.clip(np.round((g - biases[..., None]) / scales[..., None]), 0, n_bins).astype(np.uint32).reshape(N, K)
    epw = 32 // bits
    words = np.zeros((N, K // epw), np.uint32)
    for j in range(epw):
        words |= q[:, j::epw] << np.uint32(bits * j)
    return words, scales, biases


def dequantize(wq: np.ndarray, scales: np.ndarray, biases: np.ndarray, group_size: int = 64, bits: int = 4) -> np.ndarray:
    wq = np.asarray(wq, np.uint32)
    N = wq.shape[0]
    epw = 32 // bits
    K = wq.shape[1] * epw
    q = np.zeros((N, K), np.float32)
    for j in range(epw):
        q[:, j::epw] = ((wq >> np.uint32(bits * j)) & np.uint32(2 ** bits - 1)).astype(np.float32)
    s = np.repeat(np.asarray(scales, np.float32), group_size, axis=1)
    b = np.repeat(np.asarray(biases, np.float32), group_size, axis=1)
    return (s * q + b).astype(np.float32)


def quantized_matmul(x: np.ndarray, wq: np.ndarray, scales: np.ndarray, biases: np.ndarray, group_size: int = 64, bits: int = 4) -> np.ndarray:
    """mlx quantized_matmul(x, w, scales, biases, transpose=True) [3P: mlx 0.31 qmm kernels], the arithmetic behind
    QuantizedLinear (LlamaTTS.swift:958-968, Qwen3TTS.swift:1157-1170): the weight is never materialised; per group
        y[m, n] += scale[n, g] * sum_k x[m, k] q[n, k]  +  bias[n, g] * sum_k x[m, k],
    everything in float32 (x, scales, biases converted to float32 first; the caller rounds y to x's dtype).
    x [M, K] float32 (holding bf16 values), returns float32 [M, N]."""
    x = np.asarray(x, np.float32)
    wq = np.asarray(wq, np.uint32)
    N = wq.shape[0]
    epw = 32 // bits
    K = wq.shape[1] * epw
    q = np.zeros((N, K), np.float32)
    for j in range(epw):
        q[:, j::epw] = ((wq >> np.uint32(bits * j)) & np.uint32(2 ** bits - 1)).astype(np.float32)
    G = K // group_size
    s = np.asarray(scales, np.float32).reshape(N, G)
    b = np.asarray(biases, np.float32).reshape(N, G)
    xg = x.reshape(x.shape[0], G, group_size)
    qg = q.reshape(N, G, group_size)
    dots = np.einsum("mgk,ngk->mng", xg, qg).astype(np.float32)           # exact products (8-bit codes x bf16), f32 sums
    sums = xg.sum(-1, dtype=np.float32)                                    # [M, G]
    return (dots * s[None] + sums[:, None, :] * b[None]).sum(-1, dtype=np.float32)
