"""Orpheus <-> SNAC token framing on the GPU (mirrors LlamaTTS.swift:41-64 and :383-434)."""
from __future__ import annotations

import numpy as np

from . import _lib
from .generation import check


def deinterleave(codes7: np.ndarray, device: int = 0):
    """llamaDecodeAudioFromCodes' split: [B, 7G] (or [7G]) -> (l0 [B,G], l1 [B,2G], l2 [B,4G])."""
    a = np.ascontiguousarray(codes7, dtype=np.int32)
    squeeze = a.ndim == 1
    if squeeze:
        a = a[None]
    if a.shape[1] % 7:
        raise ValueError("code list length must be a multiple of 7 (parseOutput trims first, LlamaTTS.swift:424)")
    B, G = a.shape[0], a.shape[1] // 7
    l0 = np.zeros((B, G), np.int32); l1 = np.zeros((B, 2 * G), np.int32); l2 = np.zeros((B, 4 * G), np.int32)
    check(_lib.lib().mis_orpheus_deinterleave(device, a.ctypes.data, B, G, l0.ctypes.data, l1.ctypes.data, l2.ctypes.data))
    return (l0[0], l1[0], l2[0]) if squeeze else (l0, l1, l2)


def parse_output(ids: np.ndarray, lens=None, device: int = 0):
    """parseOutput per row: returns list of int32 code arrays (values = id - 128266)."""
    a = np.ascontiguousarray(ids, dtype=np.int32)
    if a.ndim == 1:
        a = a[None]
    B, S = a.shape
    lens = np.full(B, S, np.int32) if lens is None else np.ascontiguousarray(lens, dtype=np.int32)
    out = np.zeros((B, max(S, 1)), np.int32)
    n = np.zeros(B, np.int32)
    check(_lib.lib().mis_orpheus_parse_output(device, a.ctypes.data if S else None, lens.ctypes.data, B, S,
                                              out.ctypes.data if S else None, n.ctypes.data))
    return [out[b, : n[b]].copy() for b in range(B)]
