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
