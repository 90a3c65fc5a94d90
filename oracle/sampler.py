"""Logit processing + sampling of the Orpheus generate loop.  Test infrastructure only.

What the reference does (LlamaTTS.swift:691-696,717-723): `parameters.processor()` +
`parameters.sampler()` from mlx-swift-lm 3.31.4 [3P, not vendored; restated from its published
semantics, SURVEY App. C]:
  processor  RepetitionContext(penalty, contextSize): window = last `contextSize` ids of
             prompt + generated; for every id in the window (scatter => once per unique id)
                 logit = logit < 0 ? logit * pen : logit / pen
             in the logits' dtype (bf16 model => bf16 arithmetic, pen weakly typed to bf16).
  sampler    temperature == 0      -> argmax (first index on ties)
             0 < topP < 1          -> p = softmax(float32(logits)/T); sort ascending; cumsum;
                                      keep where cumsum > 1 - topP; categorical over kept p
             else                  -> categorical(logits / T)
MLX's categorical draws from MLX's global threefry stream, which cannot be reproduced outside
MLX.  Stochastic parity with the reference is therefore distributional only; what IS pinned is
(i) greedy tokens and (ii) the *set* semantics of penalty / nucleus above.  To make the engine's
own sampler testable bit-for-bit we define a deterministic realisation of those semantics,
"mis-sampler-v1", implemented identically here (numpy) and in csrc/lm_sampler.hip:

  x_i   = fdiv(l_i, T)                         IEEE float32
  y_i   = x_i - max_j x_j                      (<= 0)
  e_i   = det_exp(y_i)                         float32, fixed mul/add sequence (below), 0 if masked
  E_i   = trunc(e_i * 2^40)                    uint64 fixed point
  Z     = sum_i E_i                            exact integer => order independent
  thr   = uint64(double(float32(1) - topP) * double(Z))
  key_i = bits(e_i) >> 16                      (e truncated to bf16: ties = equal keys; MLX's own
                                                tie order inside argSort is unspecified anyway)
  k*    = min{ k : sum_{key_j <= k} E_j > thr }
  K     = { i : key_i >= k*, E_i > 0 }         (topP outside (0,1): k* = 0)
  r     = mulhi64(rand64(seed,row,step), Z_K)  Z_K = sum_{i in K} E_i
  token = first i in K (index order) whose running sum of E exceeds r      (inverse CDF, exact)
  rand64(seed,row,step) = splitmix64(splitmix64(seed ^ 0xD1B54A32D192ED03*(row+1)) + step)
`row` is the GLOBAL utterance index, so results do not depend on how a batch is sharded over GPUs.
Optional frame constraint (bench / synthetic weights only): tokens outside [lo, hi) are masked.
"""
from __future__ import annotations

import numpy as np

from .synth import bf16_round, splitmix64

F = np.float32
_LOG2E = F(1.4426950408889634)
# 2^f on [0,1): degree-6 polynomial (Horner, separate IEEE mul/add, no fma)
_P = [F(1.0), F(0.6931471805599453), F(0.2402265069591007), F(0.05550410866482158),
      F(0.009618129107628477), F(0.0013333558146428443), F(0.00015403530393381608)]


def det_exp(y: np.ndarray) -> np.ndarray:
    """Deterministic float32 exp for y <= 0 (bit-reproducible on any IEEE machine):
    t = y*log2e; n = floor(t); f = t-n; p = Horner(_P, f); result = p * 2^n; 0 when n < -60."""
    y = np.asarray(y, dtype=F)
    t = (y * _LOG2E).astype(F)
    n = np.floor(t).astype(F)
    f = (t - n).astype(F)
    p = np.full(y.shape, _P[6], dtype=F)
    for c in _P[5::-1]:
        p = ((p * f).astype(F) + c).astype(F)
    ni = np.maximum(n, F(-64)).astype(np.int32)
    out = (p * np.ldexp(F(1.0), ni).astype(F)).astype(F)
    return np.where(n < F(-60), F(0.0), out).astype(F)


def rand64(seed: int, row: int, step: int) -> int:
    with np.errstate(over="ignore"):
        a = np.uint64(seed) ^ (np.uint64(0xD1B54A32D192ED03) * np.uint64(row + 1))
        s = splitmix64(np.asarray([a], np.uint64))[0]
        s = splitmix64(np.asarray([s + np.uint64(step)], np.uint64))[0]
    return int(s)


def apply_repetition_penalty(logits: np.ndarray, window, penalty: float, bf16: bool = True) -> np.ndarray:
    """RepetitionContext.process restated (see module docstring). logits [V] float32 (holding
    bf16-representable values when bf16=True).  Returns a new array."""
    out = np.array(logits, dtype=F, copy=True)
    if penalty is None or len(window) == 0:
        return out
    pen = F(penalty)
    if bf16:
        pen = bf16_round(np.asarray([pen], F))[0]
    for t in sorted(set(int(t) for t in window)):
        l = out[t]
        v = (l * pen) if l < 0 else (l / pen)
        v = F(v)
        out[t] = bf16_round(np.asarray([v], F))[0] if bf16 else v
    return out


def sample(logits: np.ndarray, temperature: float, top_p: float, seed: int, row: int, step: int,
           lo: int = 0, hi: int | None = None, return_debug: bool = False):
    """mis-sampler-v1 on ONE row of processed logits [V] (float32)."""
    l = np.asarray(logits, dtype=F)
    V = l.shape[0]
    hi = V if hi is None else hi
    allowed = np.zeros(V, bool)
    allowed[lo:hi] = True
    if temperature == 0.0:
        masked = np.where(allowed, l, F(-np.inf))
        tok = int(np.argmax(masked))
        return (tok, {}) if return_debug else tok
    x = (l / F(temperature)).astype(F)
    m = np.max(np.where(allowed, x, F(-np.inf))).astype(F)
    e = det_exp(np.minimum((x - m).astype(F), F(0.0)))
    e = np.ascontiguousarray(np.where(allowed, e, F(0.0)).astype(F))
    E = (e.astype(np.float64) * float(2 ** 40)).astype(np.uint64)
    Z = int(E.sum(dtype=np.uint64))
    key = (e.view(np.uint32) >> np.uint32(16)).astype(np.int64)
    if 0.0 < top_p < 1.0:
        thr = int(np.uint64(np.float64(F(1.0) - F(top_p)) * np.float64(Z)))
        mass = np.zeros(1 << 16, np.uint64)
        np.add.at(mass, key, E)
        cum = np.cumsum(mass, dtype=np.uint64)
        k_star = int(np.searchsorted(cum, np.uint64(thr), side="right"))    # first key with cum > thr
        k_star = min(k_star, (1 << 16) - 1)
    else:
        k_star = 0
    keep = (key >= k_star) & (E > 0)
    Ek = np.where(keep, E, np.uint64(0))
    Zk = int(Ek.sum(dtype=np.uint64))
    r = (rand64(seed, row, step) * Zk) >> 64
    pref = np.cumsum(Ek, dtype=np.uint64)
    tok = int(np.searchsorted(pref, np.uint64(r), side="right"))
    if return_debug:
        return tok, dict(Z=Z, Zk=Zk, r=r, k_star=k_star, n_keep=int(keep.sum()), keep=keep, e=e)
    return tok


class RepetitionWindow:
    """Ring of the last `size` ids (prompt tail + sampled), RepetitionContext.prompt/didSample."""

    def __init__(self, size: int, prompt):
        self.size = size
        self.ids = [int(t) for t in prompt][-size:] if size > 0 else []

    def push(self, tok: int):
        if self.size > 0:
            self.ids = (self.ids + [int(tok)])[-self.size:]
