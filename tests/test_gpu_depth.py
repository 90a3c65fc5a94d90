"""-m gpu: the BENCHMARKED model at its real depth, and the bench's prefill instantiation at its real width
(VERDICT r02 "what's weak" 1).

(a) Orpheus-3B, all 28 layers, synthetic weights, B = 2, 48 teacher-forced positions against oracle/llama.py - with the same
    comparison at 2 and 8 layers so that the growth of the error with depth is on file (profiles/rNN_parity_observed.json).
    Beside the device error the test measures the oracle's OWN noise floor at every depth: the same graph with the same bf16
    rounding points, float64 accumulation instead of float32 (two exact realisations of one specification differ because a
    different summation order flips bf16 roundings, and the flips propagate through the residual stream).  The gate is written
    against that floor: the device may sit no further from the oracle than `FLOOR_FACTOR` x what the oracle sits from itself.
(b) `mis_lm_prefill` at Orpheus-3B width, 2 layers, B = 32 with (ragged) 32-token prompts = M = 1024 rows of the batched prefill
    GEMMs `k_gemm_pf` - the instantiation that fills every KV cache in bench.py - against the oracle, and against the
    position-by-position path (MIS_PREFILL_SEQ=1)."""
import dataclasses
import gc

import numpy as np
import pytest
import torch

import mlx_audio_swift_amd as mas
from gpu_util import lm_host_config, logits_errors, record, rms
from oracle import llama as ollama

pytestmark = pytest.mark.gpu

FLOOR_FACTOR = 2.0


class _F64Oracle(ollama.LlamaOracle):
    """Same graph, same rounding points; every contraction accumulated in float64 (the noise-floor reference)."""

    def linear(self, x, w):
        return self.r((x.to(torch.float64) @ w.to(torch.float64).t()).to(torch.float32))


def _rel_rms(a, b):
    return rms(a, b) / float(np.sqrt(np.mean(np.asarray(b, np.float64) ** 2)))


@pytest.mark.parametrize("depths", [pytest.param((2, 8), id="2_and_8_layers"),
                                    pytest.param((2, 8, 28), id="full_depth_28_layers", marks=pytest.mark.slow)])
def test_orpheus_3b_full_depth_28_layers_and_error_growth(depths):
    """(the 28-layer variant needs ~170 s, most of it the CPU oracle at 3B: marked slow - run on the final tree with --runslow; the default
    set holds the device to the same floor gate at 2 and 8 layers, the growth ~ sqrt(layers) is on file from the full runs)"""
    full = ollama.LlamaConfig()                                     # ORPHEUS_3B: 28 layers
    assert full.num_hidden_layers == 28
    # (layer keys do not depend on the layer count: the weights of the deepest variant of this run serve the shallower ones - the
    # 2- and 8-layer run does not build 28 layers of a 3B model on the CPU first)
    deepest = dataclasses.replace(full, num_hidden_layers=max(depths))
    W = ollama.make_synthetic_weights(deepest, seed=4321)
    o32 = ollama.LlamaOracle(deepest, W, round="bf16")
    del W
    gc.collect()
    o64 = _F64Oracle.__new__(_F64Oracle)
    o64.__dict__.update(o32.__dict__)                               # shares the float32 weight dict
    B, T = 2, 40
    rng = np.random.default_rng(77)
    rows = [np.concatenate([[128259], rng.integers(0, 128000, T - 1)]).astype(np.int32) for _ in range(B)]
    growth = {}
    for L in depths:
        cfg = dataclasses.replace(full, num_hidden_layers=L)
        dev = mas.LlamaTTSModel.synthetic(lm_host_config(cfg), seed=4321)
        dev.lm_reset(B, 64)
        got = np.stack([dev.lm_forward(np.asarray([r[t] for r in rows], np.int32)) for t in range(T)], axis=1)   # [B, T, V]
        del dev
        gc.collect()
        ref, ref64 = [], []
        for o, dst, sel in ((o32, ref, rows), (o64, ref64, rows[:1])):      # the float64 floor on row 0 only (it re-casts every weight per call)
            o.cfg = cfg
            o.reset(len(sel))
            dst.extend(x.numpy() for x in o.forward(sel))
        e_max = e_rms = f_max = f_rms = 0.0
        last_rms = last_floor = 0.0
        for b in range(B):
            em, er, n_sure, agree = logits_errors(got[b], ref[b])
            e_max, e_rms = max(e_max, em), max(e_rms, er)
            last_rms = max(last_rms, _rel_rms(got[b][-1], ref[b][-1]))
            if b == 0:
                f_max, f_rms, _, _ = logits_errors(ref64[0], ref[0])
                last_floor = _rel_rms(ref64[0][-1], ref[0][-1])
            assert agree, (L, b)                                    # greedy token wherever the oracle's margin exceeds 2x the error
            assert n_sure > 0
        growth[L] = dict(dev_max=e_max, dev_rms=e_rms, floor_max=f_max, floor_rms=f_rms, dev_rms_last_pos=last_rms,
                         floor_rms_last_pos=last_floor)
        record(f"orpheus3b_depth_{L}_layers", layers=L, logits_max_rel=e_max, logits_rms_rel=e_rms, oracle_f64_floor_max_rel=f_max,
               oracle_f64_floor_rms_rel=f_rms, logits_rms_rel_last_pos=last_rms, floor_rms_rel_last_pos=last_floor,
               gate=f"rms: dev <= {FLOOR_FACTOR} x floor (+1e-3); max: absolute")
    for L, g in growth.items():          # (rms against the floor; the max error is one or two bf16 ulps of the largest logit on either side
        assert g["dev_rms"] <= FLOOR_FACTOR * g["floor_rms"] + 1e-3, (L, g)       # and is bounded absolutely below)
    # absolute bounds at the benchmarked depth: twice the values observed on MI355X (profiles/r03_parity_observed.json: 28 layers
    # rms 0.0327 / max 0.0148 against the oracle's own float64 floor of 0.0321 / 0.0129; 8 layers 0.0176 vs 0.0170; 2 layers 0.0062 vs
    # 0.0058 - the device sits AT the floor at every depth, and the error grows like the floor does, ~ sqrt(layers))
    if 28 in growth:
        assert growth[28]["dev_rms"] <= 0.066 and growth[28]["dev_max"] <= 0.03, growth[28]
    assert growth[8]["dev_rms"] <= 0.036 and growth[2]["dev_rms"] <= 0.013, growth
    assert growth[8]["dev_max"] <= 0.016 and growth[2]["dev_max"] <= 0.016, growth


def test_batched_prefill_at_orpheus_3b_width_b32_m1024(monkeypatch):
    """The bench's prefill: 32 rows x 32 prompt tokens = 1024 rows through k_gemm_pf (d 3072, ffn 8192, qkv 5120).  Four rows
    are shorter (left padding: inactive positions inside the M tile).  Per-row bound: rms <= 0.008 rms(ref), the per-row bound of
    every other LM test (observed: see record); max <= 0.016 max|ref|.  The position-by-position path obeys the same bounds, and the
    two device paths differ from each other by no more than either differs from the oracle."""
    cfg = ollama.LlamaConfig(num_hidden_layers=2)
    W = ollama.make_synthetic_weights(cfg, seed=4321)
    oracle = ollama.LlamaOracle(cfg, W, round="bf16")
    del W
    dev = mas.LlamaTTSModel.synthetic(lm_host_config(cfg), seed=4321)
    B = 32
    lens = [32] * 28 + [31, 17, 2, 25]
    rng = np.random.default_rng(5)
    rows = [np.concatenate([[128259], rng.integers(0, 128000, n - 1)]).astype(np.int32) for n in lens]
    nxt = rng.integers(128266, 128266 + 4096, B).astype(np.int32)
    monkeypatch.setenv("MIS_PREFILL_SEQ", "0")
    got, hid = dev.lm_prefill(rows, max_context=64, want_hidden=True)
    got2 = dev.lm_forward(nxt)
    monkeypatch.setenv("MIS_PREFILL_SEQ", "1")
    seq, hid_s = dev.lm_prefill(rows, max_context=64, want_hidden=True)
    seq2 = dev.lm_forward(nxt)
    oracle.reset(B)
    ref_all = oracle.forward([np.concatenate([r, nxt[i:i + 1]]) for i, r in enumerate(rows)],
                             logit_positions=[[len(r) - 1, len(r)] for r in rows])
    e_b, e_s, d_bs, m_b = [], [], [], []
    for b in range(B):
        ref = ref_all[b].numpy()
        for dv, sq, rf in ((got[b], seq[b], ref[0]), (got2[b], seq2[b], ref[1])):
            em, er, _, agree = logits_errors(dv[None], rf[None])
            assert agree, b
            m_b.append(em); e_b.append(er)
            e_s.append(logits_errors(sq[None], rf[None])[1])
            d_bs.append(_rel_rms(dv, sq))
    record("batched_prefill_orpheus3b_width_m1024", logits_max_rel=max(m_b), logits_rms_rel_worst=max(e_b), logits_rms_rel_mean=float(np.mean(e_b)),
           sequential_rms_rel_worst=max(e_s), sequential_rms_rel_mean=float(np.mean(e_s)), batched_vs_sequential_rms_worst=max(d_bs),
           batched_vs_sequential_rms_mean=float(np.mean(d_bs)), tol_rms_row=0.008, tol_max=0.016)
    assert max(e_b) <= 0.008 and max(m_b) <= 0.016, (max(e_b), max(m_b))
    assert max(e_s) <= 0.008, max(e_s)
    assert max(d_bs) <= 0.008, max(d_bs)
    assert np.abs(hid - hid_s).max() <= 0.016 * np.abs(hid_s).max()
