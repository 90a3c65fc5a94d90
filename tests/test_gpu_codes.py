"""-m gpu: integer framing kernels vs the oracle and the golden fixture (bit-exact)."""
import os

import numpy as np
import pytest

import mlx_audio_swift_amd as mas
from oracle import orpheus_codes as oc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("batch,groups", [(1, 1), (1, 12), (3, 96), (32, 171), (2, 0)])
def test_deinterleave_bit_exact(batch, groups):
    rng = np.random.default_rng(groups + batch)
    rows = []
    for b in range(batch):
        l0 = rng.integers(0, 4096, groups); l1 = rng.integers(0, 4096, 2 * groups); l2 = rng.integers(0, 4096, 4 * groups)
        rows.append(oc.interleave(l0, l1, l2))
    codes = np.stack(rows).reshape(batch, 7 * groups)
    l0, l1, l2 = mas.deinterleave(codes)
    for b in range(batch):
        a, c, d = oc.deinterleave(codes[b])
        assert np.array_equal(l0[b], a) and np.array_equal(l1[b], c) and np.array_equal(l2[b], d)


def test_parse_output_golden_and_ragged_batch():
    z = np.load(os.path.join(G, "orpheus_framing.npz"))
    rows = [z[f"ids{i}"] for i in range(4)]
    S = max(len(r) for r in rows)
    ids = np.zeros((4, S), np.int32)
    for i, r in enumerate(rows):
        ids[i, : len(r)] = r
    got = mas.parse_output(ids, [len(r) for r in rows])
    for i in range(4):
        assert np.array_equal(got[i], z[f"codes{i}"])
        assert np.array_equal(got[i], oc.parse_output_row(rows[i]))


def test_parse_output_edge_cases_and_long_rows():
    rng = np.random.default_rng(5)
    off, sos, eos = oc.AUDIO_TOKEN_OFFSET, oc.START_OF_SPEECH, oc.END_OF_SPEECH
    cases = [
        [], [sos], [eos], [sos, eos], [off + 1] * 7,                      # no SOS: whole row kept
        [sos] + [off + i for i in range(6)],                              # < 1 frame -> empty
        [5, sos, 7, sos] + [off + i for i in range(15)] + [eos, off + 3], # last SOS wins; EOS dropped mid-row
        list(rng.integers(off, off + 7 * 4096, 3000)) + [sos] + list(rng.integers(off, off + 7 * 4096, 1204)),
    ]
    S = max(len(c) for c in cases)
    ids = np.zeros((len(cases), max(S, 1)), np.int32)
    for i, c in enumerate(cases):
        ids[i, : len(c)] = c
    got = mas.parse_output(ids, [len(c) for c in cases])
    for i, c in enumerate(cases):
        assert np.array_equal(got[i], oc.parse_output_row(c)), i


def test_vyvotts_parse_output_with_start_of_ai_fallback():
    # Qwen3.swift:332-358 through mis_speech_parse_output (explicit token ids)
    import ctypes as C
    from mlx_audio_swift_amd import _lib
    from mlx_audio_swift_amd.generation import check
    from oracle import orpheus_codes as oc
    rng = np.random.default_rng(11)
    A, SOS, EOS, SOA = oc.VYVO_AUDIO_OFFSET, oc.VYVO_START_OF_SPEECH, oc.VYVO_END_OF_SPEECH, oc.VYVO_START_OF_AI
    audio = lambda n: list(A + rng.integers(0, 7 * 4096, n))
    rows = [
        [5, 6, SOS] + audio(15) + [EOS],                                   # normal
        [5, SOS, 9, SOS] + audio(9) + [EOS, A + 3],                        # last start-of-speech wins; EOS dropped mid-stream
        [5, SOA, 7, 8] + audio(16),                                        # fallback: crop at the first audio token after START_OF_AI
        [5, 6, 7] + audio(3),                                              # nothing to crop, fewer than 7 -> empty
        [SOA, 3, 4],                                                       # START_OF_AI but no audio token -> whole row, trimmed to 0
        [],
    ]
    stride = max(len(r) for r in rows)
    ids = np.zeros((len(rows), stride), np.int32)
    lens = np.asarray([len(r) for r in rows], np.int32)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = r
    out = np.zeros_like(ids); n = np.zeros(len(rows), np.int32)
    check(_lib.lib().mis_speech_parse_output(0, ids.ctypes.data, lens.ctypes.data, len(rows), stride, out.ctypes.data, n.ctypes.data,
                                             SOS, EOS, A, SOA))
    for i, r in enumerate(rows):
        ref = oc.parse_output_row_vyvo(r)
        assert n[i] == len(ref) and np.array_equal(out[i, :n[i]], ref), i
