"""Helpers shared by the -m gpu parity tests (all calls go through the C ABI via the host mirror)."""
import contextlib
import os

import numpy as np
import torch

import mlx_audio_swift_amd as mas
from oracle import llama as ollama
from oracle import snac as osnac


def rms(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))


def snac_pair(cfg_dict, seed=1234):
    ocfg = osnac.SnacConfig(**cfg_dict)
    W = osnac.make_synthetic_weights(ocfg, seed=seed)
    oracle = osnac.SnacOracle(ocfg, W)
    hcfg = mas.SNACConfig(**{k: getattr(ocfg, k) for k in mas.SNACConfig.__dataclass_fields__})
    dev = mas.SNAC.from_weights(hcfg, W)
    return ocfg, oracle, dev


def lm_host_config(ocfg: ollama.LlamaConfig) -> mas.LlamaTTSConfiguration:
    return mas.LlamaTTSConfiguration(
        hidden_size=ocfg.hidden_size, num_hidden_layers=ocfg.num_hidden_layers, intermediate_size=ocfg.intermediate_size,
        num_attention_heads=ocfg.num_attention_heads, num_key_value_heads=ocfg.num_key_value_heads,
        head_dim=ocfg.head_dim, rms_norm_eps=ocfg.rms_norm_eps, vocab_size=ocfg.vocab_size, rope_theta=ocfg.rope_theta,
        rope_scaling=dict(ocfg.rope_scaling) if ocfg.rope_scaling else None, tie_word_embeddings=ocfg.tie_word_embeddings,
        qk_norm=ocfg.qk_norm, rope_plain=ocfg.rope_plain, rope_ops_in_dtype=ocfg.rope_ops_in_dtype)


def lm_pair(ocfg: ollama.LlamaConfig, seed=4321, codec=None):
    W = ollama.make_synthetic_weights(ocfg, seed=seed)        # bf16 tensors
    oracle = ollama.LlamaOracle(ocfg, W, round="bf16")
    dev = mas.LlamaTTSModel.from_weights(lm_host_config(ocfg), W, codec=codec)
    return W, oracle, dev


def teacher_forced(oracle, dev, rows, max_context=64):
    """rows: list of 1-D id arrays (ragged).  Feeds them left-aligned one token per step through the
    device engine (inactive once a row is exhausted) and all at once per row through the oracle.
    Returns per-row (device_logits [L,V], oracle_logits [L,V])."""
    B = len(rows)
    Lmax = max(len(r) for r in rows)
    dev.lm_reset(B, max_context)
    got = [[] for _ in range(B)]
    for t in range(Lmax):
        ids = np.asarray([r[t] if t < len(r) else 0 for r in rows], np.int32)
        act = np.asarray([1 if t < len(r) else 0 for r in rows], np.uint8)
        lg = dev.lm_forward(ids, act)
        for b in range(B):
            if act[b]:
                got[b].append(lg[b].copy())
    oracle.reset(B)
    ref = oracle.forward(rows)
    return [(np.stack(got[b]), ref[b].numpy()) for b in range(B)]


def record(name: str, **metrics):
    """Observed parity errors, printed (pytest -s / -rP) and appended to gpurun_out/parity_observed.jsonl so that the
    tolerances written in the tests can be compared with what the hardware actually delivers (profiles/rNN_parity_observed.json)."""
    import json
    import os
    row = {"test": name, **{k: (float(v) if isinstance(v, (int, float, np.floating, np.integer)) else v) for k, v in metrics.items()}}
    print("PARITY", json.dumps(row))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_observed.jsonl"), "a") as f:
            f.write(json.dumps(row) + "\n")
    except OSError:
        pass


def observe(kind: str, value: float, tol: float) -> bool:
    """Record one observed error of the running test next to its gate (name from PYTEST_CURRENT_TEST) and return value <= tol.
    tools/parity_summary.py reduces the records to the per-test maxima kept under profiles/."""
    name = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::")[-1]
    record(name, kind=kind, value=float(value), tol=float(tol))
    return bool(value <= tol)


def logits_errors(dev_l, ref_l):
    """(max abs err / max |ref|, rms err / rms ref, greedy agreement on rows whose oracle top-2 margin exceeds 2x the max error)"""
    scale = float(np.abs(ref_l).max())
    err = float(np.abs(dev_l - ref_l).max())
    r = rms(dev_l, ref_l) / float(np.sqrt(np.mean(ref_l.astype(np.float64) ** 2)))
    top2 = np.sort(ref_l, axis=-1)[..., -2:]
    sure = (top2[..., 1] - top2[..., 0]) > 2 * err
    agree = bool(np.array_equal(dev_l.argmax(-1)[sure], ref_l.argmax(-1)[sure]))
    return err / scale, r, int(sure.sum()), agree


@contextlib.contextmanager
def codec_exact_f32():
    """Run the enclosed codec calls on the exact-f32 kernels only (MIS_CODEC_EXACT_F32 is read per launch by csrc/codec_bf3.hip)."""
    old = os.environ.get("MIS_CODEC_EXACT_F32")
    os.environ["MIS_CODEC_EXACT_F32"] = "1"
    try:
        yield
    finally:
        if old is None:
            os.environ.pop("MIS_CODEC_EXACT_F32", None)
        else:
            os.environ["MIS_CODEC_EXACT_F32"] = old
