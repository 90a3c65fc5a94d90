"""Row a1 on the CPU tier: the host-side integer framing of the Orpheus prompt - prepareInputIds, both branches, and the
llamaEncodeAudioToCodes interleave (LlamaTTS.swift:72-98, 446-553).  No device, no SNAC encoder: the framing functions take the
reference recording's code levels as integers.  Expected values are written out from the reference's order of concatenation below;
the interleave is also checked against the oracle's pair (oracle/orpheus_codes.py, itself pinned on the de-interleave of :41-64)."""
import numpy as np
import pytest

import mlx_audio_swift_amd as mas
from oracle import orpheus_codes as oc

T = mas.OrpheusTokens


class Tok:
    """stand-in for swift-transformers' tokenizer: deterministic, length = len(text)"""
    def encode(self, s):
        return [1000 + (ord(ch) % 97) for ch in s]


def reference_prepare_input_ids(prompts, voice, ref_codes, ref_text, tok):
    """LlamaTTS.swift:446-553 restated step by step (pad FIRST, then the reference block, then [SOH] prompt [EOT][EOH]; pad length
    from the PROMPT lengths; mask = ids != pad)."""
    audio_ids = transcript = None
    if ref_codes is not None and ref_text is not None:
        audio_ids = [int(c) + T.audio_token_offset for c in ref_codes]                      # :466-467
        transcript = tok.encode(ref_text)                                                   # :468
    if voice is not None:
        prompts = [f"{voice}: {p}" for p in prompts]                                        # :474-476
    enc = [tok.encode(p) for p in prompts]
    max_len = max((len(e) for e in enc), default=0)
    rows = []
    for e in enc:
        row = [T.pad_token] * (max_len - len(e))                                            # :503-508
        if audio_ids is not None:
            row += [T.start_of_human] + transcript + [T.end_of_text, T.end_of_human]        # :521-528
            row += [T.audio_start, T.start_of_speech] + audio_ids + [T.end_of_speech, T.audio_end]
        row += [T.start_of_human] + e + [T.end_of_text, T.end_of_human]                     # :533-538
        rows.append(row)
    ids = np.asarray(rows, np.int32)
    return ids, ids != T.pad_token


def test_token_constants():
    # LlamaTTS.swift:20-30
    assert (T.start_of_human, T.end_of_human, T.end_of_text) == (128259, 128260, 128009)
    assert (T.start_of_speech, T.end_of_speech, T.pad_token) == (128257, 128258, 128263)
    assert (T.audio_start, T.audio_end, T.audio_token_offset) == (128261, 128262, 128266)


@pytest.mark.parametrize("voice", [None, "tara"])
def test_plain_branch_matches_the_reference_order(voice):
    prompts = ["hi", "a longer prompt, so the first row is padded", ""]
    rows = mas.orpheus_prompt_rows(Tok(), prompts, voice)
    assert [r.dtype for r in rows] == [np.int32] * 3
    for p, r in zip(prompts, rows):
        text = p if voice is None else f"{voice}: {p}"
        assert r.tolist() == [T.start_of_human] + Tok().encode(text) + [T.end_of_text, T.end_of_human]
    ids, mask = mas.padded_prompt_batch(rows)
    want_ids, want_mask = reference_prepare_input_ids(prompts, voice, None, None, Tok())
    assert np.array_equal(ids, want_ids) and np.array_equal(mask, want_mask)
    assert mask[1].all() and not mask[0, 0] and mask[0, -1]                      # left padding


def test_voice_cloning_branch_matches_the_reference_order():
    rng = np.random.default_rng(11)
    g = 5
    l1, l2, l3 = (rng.integers(0, 4096, n * g) for n in (1, 2, 4))
    codes = mas.interleave_snac_codes(l1, l2, l3)
    prompts = ["first", "second prompt"]
    rows = mas.orpheus_prompt_rows(Tok(), prompts, "leo", codes, "what was said")
    head = ([T.start_of_human] + Tok().encode("what was said") + [T.end_of_text, T.end_of_human, T.audio_start, T.start_of_speech] +
            [int(c) + T.audio_token_offset for c in codes] + [T.end_of_speech, T.audio_end])
    for p, r in zip(prompts, rows):
        assert r.tolist() == head + [T.start_of_human] + Tok().encode(f"leo: {p}") + [T.end_of_text, T.end_of_human]
    ids, mask = mas.padded_prompt_batch(rows)
    want_ids, want_mask = reference_prepare_input_ids(prompts, "leo", codes, "what was said", Tok())
    assert np.array_equal(ids, want_ids) and np.array_equal(mask, want_mask)
    # the reference block needs BOTH the recording and its transcript (`if let refAudio, let refText`, :457)
    for rc, rt in ((codes, None), (None, "what was said")):
        plain = mas.orpheus_prompt_rows(Tok(), prompts, "leo", rc, rt)
        assert all(a.tolist() == b.tolist() for a, b in zip(plain, mas.orpheus_prompt_rows(Tok(), prompts, "leo")))
    # every audio id of the block lies in its frame slot's range, so parseOutput / the frame-constrained sampler read it back
    block = np.asarray(rows[0][len(head) - 2 - len(codes):len(head) - 2]) - T.audio_token_offset
    assert np.array_equal(block, codes) and np.array_equal(block // 4096, np.tile(np.arange(7), g))


def test_interleave_is_the_inverse_of_the_decode_side_framing():
    rng = np.random.default_rng(3)
    for g in (1, 2, 12):
        l1, l2, l3 = (rng.integers(0, 4096, n * g) for n in (1, 2, 4))
        flat = mas.interleave_snac_codes(l1, l2, l3)
        assert flat.dtype == np.int32 and flat.shape == (7 * g,)
        assert np.array_equal(flat, oc.interleave(l1, l2, l3))
        a, b, c = oc.deinterleave(flat)
        assert np.array_equal(a, l1) and np.array_equal(b, l2) and np.array_equal(c, l3)
    assert mas.interleave_snac_codes([], [], []).shape == (0,)
    with pytest.raises(mas.AudioGenerationError):
        mas.interleave_snac_codes([1, 2], [1, 2, 3], [1] * 8)


def test_empty_batch_and_ragged_rows():
    ids, mask = mas.padded_prompt_batch([])
    assert ids.shape == (0, 0) and mask.shape == (0, 0)
    ids, mask = mas.padded_prompt_batch([np.asarray([5, 6, 7], np.int32), np.asarray([], np.int32)])
    assert ids.tolist() == [[5, 6, 7], [T.pad_token] * 3] and mask.tolist() == [[True] * 3, [False] * 3]
