"""Oracle pinning for the integer framing (reference: LlamaTTS.swift:20-98,383-434)."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import orpheus_codes as oc


def test_token_constants():
    # OrpheusTokens, LlamaTTS.swift:20-30
    assert (oc.START_OF_HUMAN, oc.END_OF_HUMAN, oc.END_OF_TEXT) == (128259, 128260, 128009)
    assert (oc.START_OF_SPEECH, oc.END_OF_SPEECH, oc.PAD_TOKEN) == (128257, 128258, 128263)
    assert (oc.AUDIO_START, oc.AUDIO_END, oc.AUDIO_TOKEN_OFFSET) == (128261, 128262, 128266)


def test_deinterleave_known_frame():
    # one frame, slot k carries value k*4096 + (k+1): LlamaTTS.swift:51-57 ordering L0,L1,L2,L2,L1,L2,L2
    frame = [k * 4096 + (k + 1) for k in range(7)]
    l1, l2, l3 = oc.deinterleave(frame)
    assert l1.tolist() == [1]
    assert l2.tolist() == [2, 5]
    assert l3.tolist() == [3, 4, 6, 7]


@settings(max_examples=50, deadline=None)
@given(st.integers(0, 40), st.integers(0, 2 ** 31 - 1))
def test_interleave_is_exact_inverse(groups, seed):
    # LlamaTTS.swift:72-98 must invert :41-69
    rng = np.random.default_rng(seed)
    l1 = rng.integers(0, 4096, groups, dtype=np.int32)
    l2 = rng.integers(0, 4096, 2 * groups, dtype=np.int32)
    l3 = rng.integers(0, 4096, 4 * groups, dtype=np.int32)
    codes = oc.interleave(l1, l2, l3)
    assert codes.shape == (7 * groups,)
    if groups:
        assert codes.min() >= 0 and codes.max() < 7 * 4096
    a, b, c = oc.deinterleave(codes)
    assert np.array_equal(a, l1) and np.array_equal(b, l2) and np.array_equal(c, l3)


def test_parse_output_crops_after_last_start_of_speech_and_trims():
    off = oc.AUDIO_TOKEN_OFFSET
    ids = [oc.START_OF_HUMAN, 5, 6, oc.END_OF_TEXT, oc.END_OF_HUMAN, oc.START_OF_SPEECH, 9, 9,
           oc.START_OF_SPEECH] + [off + i for i in range(16)] + [oc.END_OF_SPEECH]
    out = oc.parse_output_row(ids)
    assert out.tolist() == list(range(14))          # 16 -> trimmed to 14 (:424), EOS dropped (:415)


def test_parse_output_without_start_of_speech_keeps_everything():
    # lastOccurrenceIdx == nil -> whole row (LlamaTTS.swift:401-405), prompt included (App. D.4)
    off = oc.AUDIO_TOKEN_OFFSET
    ids = [off + 1] * 7
    assert oc.parse_output_row(ids).tolist() == [1] * 7
    assert oc.parse_output_row([]).tolist() == []
    assert oc.parse_output_row([oc.START_OF_SPEECH]).tolist() == []


def test_wrap_and_left_pad():
    a = oc.wrap_prompt([1, 2, 3])
    assert a.tolist() == [128259, 1, 2, 3, 128009, 128260]
    ids, mask = oc.left_pad_batch([a, oc.wrap_prompt([7])])
    assert ids.shape == (2, 6)
    assert ids[1].tolist() == [128263, 128263, 128259, 7, 128009, 128260]
    assert mask[1].tolist() == [False, False, True, True, True, True]


def test_vyvotts_chunked_decode_is_a_concatenation_of_independent_decodes():
    """decodeAudioFromCodes (Qwen3.swift:47-83): <= chunk groups -> one decode; else independent chunk decodes, concatenated."""
    from oracle import snac as osnac
    cfg = osnac.SnacConfig(**osnac.TINY)
    so = osnac.SnacOracle(cfg, osnac.make_synthetic_weights(cfg, seed=5))
    rng = np.random.default_rng(4)
    groups = 7
    codes = []
    for _ in range(groups):
        codes += [int(rng.integers(0, cfg.codebook_size)) + k * 4096 for k in range(7)]
    hop = 1
    for r in cfg.decoder_rates:
        hop *= r
    hop *= cfg.vq_strides[0]
    whole = oc.decode_audio_from_codes_chunked(codes, so, chunk_groups=50)
    l0, l1, l2 = oc.deinterleave(codes)
    assert np.array_equal(whole, so.decode([l0[None], l1[None], l2[None]], None)[0, 0]) and len(whole) == groups * hop
    ch = oc.decode_audio_from_codes_chunked(codes, so, chunk_groups=3)          # chunks of 3, 3, 1 groups
    assert len(ch) == groups * hop
    for a, b in ((0, 3), (3, 6), (6, 7)):
        m0, m1, m2 = oc.deinterleave(codes[7 * a: 7 * b])
        assert np.array_equal(ch[a * hop: b * hop], so.decode([m0[None], m1[None], m2[None]], None)[0, 0])
    assert not np.allclose(ch, whole, atol=1e-6)                                # nothing is carried across the boundaries
    # explicit noise is sliced per chunk
    nz = [rng.standard_normal((1, n)).astype(np.float32) for n in so.noise_lengths(groups)]
    chn = oc.decode_audio_from_codes_chunked(codes, so, chunk_groups=3, noises=nz)
    m0, m1, m2 = oc.deinterleave(codes[21:42])
    per = [z.shape[-1] // groups for z in nz]
    ref = so.decode([m0[None], m1[None], m2[None]], [z[..., 3 * p_: 6 * p_] for z, p_ in zip(nz, per)])[0, 0]
    assert np.array_equal(chn[3 * hop: 6 * hop], ref)
