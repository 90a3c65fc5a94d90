"""CPU emulation of the split-bf16 contraction (x = xh + xl, w = wh + wl with bf16 halves; y = xh.wh + xh.wl + xl.wh accumulated in
f32 -- the lo.lo term dropped) inside the SNAC oracle's dense convolutions, against the plain f32 oracle: the waveform error the
3-MFMA bf16 path would add, measured before any kernel is written.  Depthwise convs / Snake / noise stay f32 as on the device.
Output: one JSON line (also kept as profiles/r02_bf16x3_emulation.json)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import snac as osn


def bf16_round(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def split(a):
    h = bf16_round(a)
    return h, bf16_round(a - h)


_einsum = np.einsum
TERMS = 3


def einsum_split(spec, a, b):
    ah, al = split(a)
    bh, bl = split(b)
    y = _einsum(spec, ah, bh) + _einsum(spec, ah, bl) + _einsum(spec, al, bh)
    if TERMS == 4:
        y = y + _einsum(spec, al, bl)
    if TERMS == 1:
        y = _einsum(spec, ah, bh)
    return y.astype(np.float32)


def run(cfg, groups, batch, seed=1234):
    W = osn.make_synthetic_weights(cfg, seed=seed)
    orc = osn.SnacOracle(cfg, W)
    codes = osn.synthetic_codes(cfg, batch, groups)
    noise = osn.synthetic_noise(cfg, batch, groups)
    ref = orc.decode(codes, noise)
    out = {}
    global TERMS
    for terms in (3, 4, 1):
        TERMS = terms
        np.einsum = einsum_split
        try:
            got = orc.decode(codes, noise)
        finally:
            np.einsum = _einsum
        d = (got - ref).astype(np.float64)
        out[f"terms{terms}"] = {"rms_err": float(np.sqrt((d * d).mean())), "max_err": float(np.abs(d).max()),
                                "rel_rms": float(np.sqrt((d * d).mean()) / np.sqrt((ref.astype(np.float64) ** 2).mean()))}
    out["ref_rms"] = float(np.sqrt((ref.astype(np.float64) ** 2).mean()))
    return out


if __name__ == "__main__":
    cfg = osn.SnacConfig()                        # the 24 kHz configuration (C1)
    res = {"what": "SNAC 24 kHz decode, 12 groups, batch 1, synthetic weights: split-bf16 dense convs vs f32 oracle",
           "gate": "north_star: within 1e-4 RMS waveform error", "c1": run(cfg, 12, 1)}
    print(json.dumps(res))
