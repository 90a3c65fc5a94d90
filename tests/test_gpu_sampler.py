"""-m gpu: on-device logit processor + sampler vs the numpy oracle of the same spec (bit-exact)."""
import numpy as np
import pytest

import mlx_audio_swift_amd as mas
from mlx_audio_swift_amd.tts import sample_logits
from oracle import sampler as osamp
from oracle.synth import bf16_round

pytestmark = pytest.mark.gpu


def _oracle_tokens(logits, window, wl, p, step, lo=0, hi=None):
    out = []
    for b in range(logits.shape[0]):
        win = window[b, window.shape[1] - wl[b]:] if window.shape[1] else []
        l = osamp.apply_repetition_penalty(logits[b], win, p.repetition_penalty, bf16=True) \
            if p.repetition_penalty and p.repetition_penalty != 1.0 else logits[b]
        out.append(osamp.sample(l, p.temperature, p.top_p, p.seed, p.row_offset + b, step, lo, hi))
    return np.asarray(out, np.int32)


@pytest.mark.parametrize("V,temp,top_p,pen", [(1448, 0.6, 0.8, 1.3), (1448, 0.0, 0.8, 1.3), (5000, 1.0, 1.0, 0.0),
                                               (156940, 0.6, 0.8, 1.3), (156940, 0.9, 0.3, 1.1)])
def test_sampler_bit_exact_vs_oracle(V, temp, top_p, pen):
    rng = np.random.default_rng(V + int(temp * 10))
    B, ctx = 6, 20
    logits = bf16_round((rng.standard_normal((B, V)) * 2.5).astype(np.float32))
    window = rng.integers(0, V, (B, ctx)).astype(np.int32)
    window[:, -3] = window[:, -1]                               # duplicate ids in the window
    wl = np.asarray([20, 20, 7, 0, 1, 20], np.int32)
    p = mas.GenerateParameters(temperature=temp, top_p=top_p, repetition_penalty=pen, seed=1234, row_offset=10)
    for step in (0, 5):
        got = sample_logits(logits, window, wl, p, step)
        ref = _oracle_tokens(logits, window, wl, p, step)
        assert np.array_equal(got, ref), (step, got, ref)


def test_range_constraint_and_peaked_distribution():
    rng = np.random.default_rng(9)
    V = 156940
    logits = bf16_round((rng.standard_normal((3, V)) * 1.0).astype(np.float32))
    logits[0, 130000] = 30.0                                    # peaked row: nucleus = 1 token
    p = mas.GenerateParameters(temperature=0.6, top_p=0.8, repetition_penalty=0.0, seed=7)
    got = sample_logits(logits, np.zeros((3, 0), np.int32), np.zeros(3, np.int32), p, 3)
    assert got[0] == 130000
    assert np.array_equal(got, _oracle_tokens(logits, np.zeros((3, 0), np.int32), np.zeros(3, np.int32), p, 3))
    lo, hi = 128266 + 2 * 4096, 128266 + 3 * 4096
    got = sample_logits(logits, np.zeros((3, 0), np.int32), np.zeros(3, np.int32), p, 3, lo, hi)
    assert np.all((got >= lo) & (got < hi))
    assert np.array_equal(got, _oracle_tokens(logits, np.zeros((3, 0), np.int32), np.zeros(3, np.int32), p, 3, lo, hi))
    pc = mas.GenerateParameters(temperature=0.6, top_p=0.8, repetition_penalty=0.0, seed=7, frame_constrained=True)
    for step in (0, 1, 6, 7, 13):
        g = sample_logits(logits, np.zeros((3, 0), np.int32), np.zeros(3, np.int32), pc, step)
        s = step % 7
        assert np.all((g >= 128266 + s * 4096) & (g < 128266 + (s + 1) * 4096))


@pytest.mark.parametrize("temp,top_p,pen", [(0.6, 0.8, 1.3), (0.0, 0.8, 1.3), (1.0, 1.0, 0.0), (0.9, 0.3, 1.1), (0.6, 0.999, 1.3)])
def test_single_launch_sampler_on_narrow_ranges_bit_exact(temp, top_p, pen):
    """Allowed ranges of <= 4096 ids (every frame-constrained step, the bench's case) run k_samp_narrow: one launch instead of
    six, same integers.  Frame slots 0..6 of two frames, penalty windows holding ids inside AND outside the allowed range,
    duplicate ids, a ragged range [lo, hi) whose width is not a multiple of 4, and a range clipped by the vocabulary end."""
    rng = np.random.default_rng(int(temp * 100) + int(top_p * 1000))
    V, B, ctx = 156940, 6, 24
    logits = bf16_round((rng.standard_normal((B, V)) * 2.5).astype(np.float32))
    wl = np.asarray([24, 24, 7, 0, 1, 24], np.int32)
    pc = mas.GenerateParameters(temperature=temp, top_p=top_p, repetition_penalty=pen, seed=99, row_offset=3, frame_constrained=True)
    for step in (0, 1, 5, 6, 7, 13):
        lo = 128266 + (step % 7) * 4096
        window = rng.integers(lo, lo + 4096, (B, ctx)).astype(np.int32)
        window[:, ::5] = rng.integers(0, V, (B, len(range(0, ctx, 5))))        # some ids outside the range
        window[:, -3] = window[:, -1]
        # the strongest candidates sit in the window so the penalty decides the outcome
        for b in range(B):
            logits[b, window[b, -1]] = 9.0
            logits[b, window[b, -2]] = 8.5
        logits = bf16_round(logits)
        got = sample_logits(logits, window, wl, pc, step)
        ref = _oracle_tokens(logits, window, wl, pc, step, lo, lo + 4096)
        assert np.array_equal(got, ref), (step, got, ref)
    p = mas.GenerateParameters(temperature=temp, top_p=top_p, repetition_penalty=pen, seed=5, row_offset=0)
    window = rng.integers(0, V, (B, ctx)).astype(np.int32)
    for lo, hi in ((1000, 1000 + 1447), (V - 301, V), (70000, 70001)):
        got = sample_logits(logits, window, wl, p, 2, lo, hi)
        ref = _oracle_tokens(logits, window, wl, p, 2, lo, hi)
        assert np.array_equal(got, ref), (lo, hi, got, ref)


@pytest.mark.parametrize("temp,top_p,pen", [(0.6, 0.8, 1.3), (0.0, 0.8, 1.3), (1.0, 1.0, 0.0), (0.9, 0.3, 1.1), (0.7, 0.95, 1.5)])
def test_full_vocabulary_single_launch_sampler_bit_exact(temp, top_p, pen, monkeypatch):
    """The full 156 940-id range in ONE launch (k_samp_cluster: 8 blocks per row, four row-local barriers) against the oracle and
    against the six-kernel path (MIS_SAMPLER_WIDE=1): same integers, same token.  32 rows = the bench's batch (256 co-resident
    blocks); penalty windows with duplicates and with the row maximum inside; wide [lo, hi) ranges; repeated calls on the same
    scratch (the exchange area must come back zeroed)."""
    rng = np.random.default_rng(int(temp * 100) + int(top_p * 1000) + 5)
    V, B, ctx = 156940, 32, 20
    logits = bf16_round((rng.standard_normal((B, V)) * 2.0).astype(np.float32))
    window = rng.integers(0, V, (B, ctx)).astype(np.int32)
    window[:, -3] = window[:, -1]
    for b in range(B):
        window[b, 0] = int(np.argmax(logits[b]))                  # the maximum is penalised: the row max must follow
    wl = np.asarray([20, 20, 7, 0, 1, 20] * 5 + [20, 3], np.int32)
    p = mas.GenerateParameters(temperature=temp, top_p=top_p, repetition_penalty=pen, seed=4321, row_offset=7)
    for step, (lo, hi) in ((0, (0, None)), (3, (0, None)), (9, (1000, 150001)), (10, (128266, 128266 + 7 * 4096))):
        monkeypatch.setenv("MIS_SAMPLER_WIDE", "0")
        got = sample_logits(logits, window, wl, p, step, lo, hi if hi is not None else 0)
        monkeypatch.setenv("MIS_SAMPLER_WIDE", "1")
        six = sample_logits(logits, window, wl, p, step, lo, hi if hi is not None else 0)
        ref = _oracle_tokens(logits, window, wl, p, step, lo, hi)
        assert np.array_equal(got, ref), (step, np.nonzero(got != ref)[0], got[:8], ref[:8])
        assert np.array_equal(six, ref), step


def test_one_launch_sampler_timeout_falls_back_to_the_multi_launch_path(monkeypatch):
    """The one-launch sampler's 8 blocks of a row wait for each other; MIS_SAMPLER_SPIN=0 lets a block give up without a single poll,
    which is what a row whose partner blocks are not resident (another stream holding the CUs) looks like.  The failed rows raise
    c_fail; the stand-alone entry point owns its inputs and falls back (fresh logits and windows, re-initialised exchange area,
    six-kernel path): the tokens are still the oracle's.  A later call on the default limit works again (nothing is left dirty)."""
    rng = np.random.default_rng(41)
    V, B, ctx = 156940, 32, 20
    logits = bf16_round((rng.standard_normal((B, V)) * 2.0).astype(np.float32))
    window = rng.integers(0, V, (B, ctx)).astype(np.int32)
    wl = np.full(B, 20, np.int32)
    p = mas.GenerateParameters(temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=99, row_offset=3)
    ref = _oracle_tokens(logits, window, wl, p, 4)
    monkeypatch.setenv("MIS_SAMPLER_WIDE", "0")
    monkeypatch.setenv("MIS_SAMPLER_SPIN", "0")
    got = sample_logits(logits, window, wl, p, 4)
    assert np.array_equal(got, ref)
    monkeypatch.delenv("MIS_SAMPLER_SPIN")
    assert np.array_equal(sample_logits(logits, window, wl, p, 4), ref)
