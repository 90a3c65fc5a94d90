"""-m gpu: SNAC decode (HIP, through the C ABI) vs the CPU oracle.
Tolerance: waveform RMS error <= 1e-4 (BASELINE.json north_star), measured on outputs in (-1,1)."""
import os

import numpy as np
import pytest

from gpu_util import rms, snac_pair
from oracle import snac as osnac

pytestmark = pytest.mark.gpu
TOL = 1e-4
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("cfgd,batch,groups", [
    (osnac.TINY, 2, 5), (osnac.TINY, 1, 1), (osnac.TINY, 3, 40),
    (dict(osnac.TINY, decoder_rates=[3, 2], decoder_dim=48), 2, 7),          # odd stride: length s*T-1
    (dict(osnac.TINY, vq_strides=[2, 1], decoder_dim=128, codebook_size=4096), 1, 9),
])
def test_decode_matches_oracle_small(cfgd, batch, groups):
    ocfg, oracle, dev = snac_pair(cfgd)
    codes = osnac.synthetic_codes(ocfg, batch, groups)
    noise = osnac.synthetic_noise(ocfg, batch, groups)
    ref, inter = oracle.decoder(oracle.from_codes(codes), noise, return_intermediates=True)
    got = dev.decode(codes, noise)
    assert got.shape == ref.shape
    # stage-by-stage diagnostics first (relative to the stage's own scale)
    zq = dev.debug_tap("zq", batch)
    assert rms(zq, oracle.from_codes(codes)) < 1e-5 * max(1.0, float(np.abs(zq).max()))
    for name in ["stem_dw", "stem_pw"] + [f"block{i}" for i in range(len(ocfg.decoder_rates))]:
        t = dev.debug_tap(name, batch)
        assert t.shape == inter[name].shape, name
        assert rms(t, inter[name]) < 2e-5 * float(np.abs(inter[name]).max()), name
    assert rms(got, ref) < TOL
    assert np.all(np.isfinite(got)) and np.abs(got).max() < 1.0


def test_c1_full_24khz_matches_golden_and_oracle():
    # BASELINE config 1: SNAC 24 kHz, 1 s of random codebook indices (12 groups -> 24576 samples)
    z = np.load(os.path.join(G, "snac_c1.npz"))
    ocfg, oracle, dev = snac_pair({})
    codes = [z["l0"], z["l1"], z["l2"]]
    noise = osnac.synthetic_noise(ocfg, 1, 12, seed=1236)
    got = dev.decode(codes, noise)
    assert got.shape == (1, 1, 24576)
    assert rms(got[0, 0], z["pcm_noise"]) < TOL
    dev.set_noise(True)
    got0 = dev.decode(codes, None)
    assert rms(got0[0, 0], z["pcm_zero_noise"]) < TOL
    zeros = [np.zeros((1, n), np.float32) for n in dev.noise_lengths(12)]
    assert np.array_equal(dev.decode(codes, zeros), got0)           # explicit zeros == "add nothing"
    assert dev.noise_lengths(12) == [384, 3072, 12288, 24576] and dev.num_samples(12) == 24576


def test_full_config_batch_and_internal_noise():
    ocfg, oracle, dev = snac_pair({})
    B, Gr = 3, 4
    codes = osnac.synthetic_codes(ocfg, B, Gr, seed=77)
    noise = osnac.synthetic_noise(ocfg, B, Gr, seed=78)
    got = dev.decode(codes, noise)
    assert rms(got, oracle.decode(codes, noise)) < TOL
    # batch-vs-single parity (pattern of Tests/ParakeetBatchParityTests.swift): row r == the B=1 decode
    for r in range(B):
        one = dev.decode([c[r:r + 1] for c in codes], [n[r:r + 1] for n in noise])
        assert np.array_equal(one[0], got[r])
    # noise == None: N(0,1) drawn on the device (reference behaviour): reproducible per seed, differs per seed
    dev.set_noise(False, seed=1)
    a = dev.decode(codes, None); b = dev.decode(codes, None)
    dev.set_noise(False, seed=2)
    c = dev.decode(codes, None)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    dev.set_noise(True)
    d = dev.decode(codes, None)
    assert 0.01 < rms(a, d) < 1.0                                   # noise really is injected


def test_large_batch_properties_at_bench_size():
    # BASELINE config 3 codec leg: B = 32 rows x 96 groups (8.192 s each); the oracle is too slow here,
    # so check size-independent properties: rows independent, finite, bounded, deterministic.
    ocfg, oracle, dev = snac_pair({})
    dev.set_noise(True)
    B, Gr = 32, 96
    codes = osnac.synthetic_codes(ocfg, B, Gr, seed=5)
    got = dev.decode(codes, None)
    assert got.shape == (B, 1, Gr * 2048)
    assert np.all(np.isfinite(got)) and np.abs(got).max() < 1.0 and got.std() > 0.05
    for r in (0, 17, 31):
        one = dev.decode([c[r:r + 1] for c in codes], None)
        assert np.array_equal(one[0], got[r])
    # prefix property: the first groups of a row do not depend on codes far to the right
    # (receptive field of the decoder is bounded): decode the first 8 groups alone
    head = dev.decode([codes[0][:1, :8], codes[1][:1, :16], codes[2][:1, :32]], None)
    assert rms(head[0, 0, : 4 * 2048], got[0, 0, : 4 * 2048]) < 1e-6


def test_empty_and_invalid_inputs():
    import mlx_audio_swift_amd as mas
    ocfg, oracle, dev = snac_pair(osnac.TINY)
    out = dev.decode([np.zeros((2, 0), np.int32)] * 3)
    assert out.shape == (2, 1, 0)
    with pytest.raises(mas.AudioGenerationError):
        dev.decode([np.zeros((1, 4), np.int32), np.zeros((1, 7), np.int32), np.zeros((1, 16), np.int32)])
    with pytest.raises(mas.AudioGenerationError) as e:
        m = mas.SNAC(mas.SNACConfig(**{k: getattr(ocfg, k) for k in mas.SNACConfig.__dataclass_fields__}))
        m.finalize()                                               # verify:.all -> missing keys
    assert e.value.case == "modelNotInitialized"


def test_encode_matches_oracle():
    # (f)2: SNAC.encode (SNACDecoder.swift:120-125): encoder latent within fp tolerance, codes equal wherever the oracle's
    # nearest-code decision has a margin (a tie within float rounding may legitimately go either way)
    import mlx_audio_swift_amd as mas
    from oracle import snac as osnac
    ocfg = osnac.SnacConfig(**osnac.TINY)
    W = osnac.make_synthetic_weights(ocfg, with_encoder=True)
    orc = osnac.SnacOracle(ocfg, W)
    hcfg = mas.SNACConfig(**{k: getattr(ocfg, k) for k in mas.SNACConfig.__dataclass_fields__})
    dev = mas.SNAC.from_weights(hcfg, W)
    rng = np.random.default_rng(4)
    for B, n in ((2, 1000), (1, 64), (3, 4133)):
        audio = (0.3 * rng.standard_normal((B, n))).astype(np.float32)
        assert dev.padded_length(n) == orc.preprocess(audio[:, None]).shape[-1]
        codes, z = dev.encode(audio, return_latent=True)
        zr = orc.encoder(orc.preprocess(audio[:, None]))
        assert z.shape == zr.shape and np.abs(z - zr).max() <= 2e-4 * np.abs(zr).max()
        rcodes, dist = orc.encode(audio[:, None], return_details=True)
        for lvl, (g, r, d) in enumerate(zip(codes, rcodes, dist)):
            assert g.shape == r.shape
            ds = np.sort(d, -1)
            sure = (ds[..., 1] - ds[..., 0]) > 1e-4
            assert np.array_equal(g[sure], r[sure]), lvl
            # where they differ the engine's choice is as close as the oracle's
            pick = np.take_along_axis(d, g[..., None].astype(np.int64), -1)[..., 0]
            assert np.all(pick <= ds[..., 0] + 1e-4)
            if lvl == 0:
                assert sure.mean() > 0.9
    # round trip through the engine's own decoder runs (AudioCodecModel: decodeAudio(encodeAudio(x)))
    wav = dev.decode(dev.encode_audio((0.3 * rng.standard_normal((1, 2048))).astype(np.float32)), noise=None)
    assert wav.shape[-1] == 2048 // 16 * 32          # TINY: encoder hop 16, decoder hop 32
    # a handle without encoder tensors raises audioEncodingFailed
    dec_only = mas.SNAC.from_weights(hcfg, osnac.make_synthetic_weights(ocfg))
    with pytest.raises(mas.AudioGenerationError) as e:
        dec_only.encode(np.zeros((1, 64), np.float32))
    assert e.value.case == "audioEncodingFailed"


def test_orpheus_voice_cloning_prompt_uses_the_encode_path():
    # llamaEncodeAudioToCodes + the refAudio/refText branch of prepareInputIds (LlamaTTS.swift:72-98,457-528)
    import mlx_audio_swift_amd as mas
    from oracle import orpheus_codes as oc
    from oracle import snac as osnac
    small = dict(encoder_dim=4, encoder_rates=[2, 4, 8, 8], decoder_dim=64, decoder_rates=[8, 8, 4, 2], codebook_size=4096,
                 codebook_dim=8, vq_strides=[4, 2, 1])
    ocfg = osnac.SnacConfig(**small)
    codec = mas.SNAC.from_weights(mas.SNACConfig(**{k: getattr(ocfg, k) for k in mas.SNACConfig.__dataclass_fields__}),
                                  osnac.make_synthetic_weights(ocfg, with_encoder=True))
    lm = mas.LlamaTTSModel.synthetic(mas.LlamaTTSConfiguration(hidden_size=64, num_hidden_layers=1, intermediate_size=64,
                                                               num_attention_heads=1, num_key_value_heads=1, head_dim=64,
                                                               vocab_size=156940), codec=codec)

    class Tok:
        def encode(self, s):
            return [1000 + (ord(ch) % 97) for ch in s]
    lm.tokenizer = Tok()
    rng = np.random.default_rng(5)
    ref = (0.3 * rng.standard_normal(5000)).astype(np.float32)          # padded to 3 groups of 2048 samples
    flat = lm.encode_audio_to_codes(ref)
    levels = codec.encode(ref)
    assert flat.shape == (21,) and np.array_equal(flat, oc.interleave(levels[0][0], levels[1][0], levels[2][0]))
    l0, l1, l2 = oc.deinterleave(flat)                                   # decode-side framing is the exact inverse
    assert np.array_equal(l0, levels[0][0]) and np.array_equal(l1, levels[1][0]) and np.array_equal(l2, levels[2][0])
    row = lm.prepare_input_ids(["hi"], voice="tara", ref_audio=ref, ref_text="ok")[0]
    T = mas.OrpheusTokens
    want = ([T.start_of_human] + Tok().encode("ok") + [T.end_of_text, T.end_of_human, T.audio_start, T.start_of_speech] +
            list(flat + T.audio_token_offset) + [T.end_of_speech, T.audio_end, T.start_of_human] + Tok().encode("tara: hi") +
            [T.end_of_text, T.end_of_human])
    assert row.tolist() == want


def test_snac_encode_decode_cycle_on_the_reference_fixture():
    """Mirror of the reference's snacEncodeDecodeCycle smoke test (Tests/MLXAudioSmokeTests.swift:78-110): intention.wav (24 kHz)
    -> SNAC.encode -> SNAC.decode, full snac_24khz dimensions, synthetic seeded weights (no checkpoints offline).  The reference
    only expects a non-empty reconstruction; here the codes and the waveform are also compared with the oracle."""
    import wave
    import mlx_audio_swift_amd as mas
    from gpu_util import record
    w = wave.open(os.path.join(G, "intention.wav"))
    pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
    assert w.getframerate() == 24000
    ocfg = osnac.SnacConfig()
    W = osnac.make_synthetic_weights(ocfg, seed=1234, with_encoder=True)
    orc = osnac.SnacOracle(ocfg, W)
    hcfg = mas.SNACConfig(**{k: getattr(ocfg, k) for k in mas.SNACConfig.__dataclass_fields__})
    dev = mas.SNAC.from_weights(hcfg, W)
    audio = pcm[None]                                                    # [1, 36480]
    codes = dev.encode_audio(audio)                                      # AudioCodecModel.encodeAudio
    rcodes, dist = orc.encode(audio[:, None], return_details=True)
    padded = dev.padded_length(audio.shape[1])
    assert padded == 36864 and [c.shape for c in codes] == [(1, 18), (1, 36), (1, 72)]       # hop 512 x lcm(4,2,1): 72 latent frames
    agree = []
    for g, r, d in zip(codes, rcodes, dist):
        ds = np.sort(d, -1)
        sure = (ds[..., 1] - ds[..., 0]) > 1e-4
        assert np.array_equal(g[sure], r[sure])
        agree.append(float(np.mean(g == r)))
    noise = osnac.synthetic_noise(ocfg, 1, 18, seed=3)
    wav = dev.decode(codes, noise)                                       # AudioCodecModel.decodeAudio
    ref = orc.decode([np.asarray(c) for c in codes], noise)
    e = rms(wav, ref)
    assert wav.shape == ref.shape == (1, 1, 36864) and wav.shape[-1] > 0 and e < TOL, e
    record("intention_wav_snac_cycle", code_agreement=agree, wave_rms=e, tol_rms=TOL)


def test_local_mha_variant_decode_and_encode_match_oracle():
    """The 32 / 44 kHz model family in miniature (oracle TINY_ATTN): LocalMHA (LayerNorm, to_qkv, rotary windowed attention,
    to_out + residual; Attention.swift:14-185) after the decoder stem and before the encoder's last conv, four codebooks."""
    import mlx_audio_swift_amd as mas
    ocfg = osnac.SnacConfig(**osnac.TINY_ATTN)
    W = osnac.make_synthetic_weights(ocfg, with_encoder=True)
    orc = osnac.SnacOracle(ocfg, W)
    hcfg = mas.SNACConfig(**{k: getattr(ocfg, k) for k in mas.SNACConfig.__dataclass_fields__})
    dev = mas.SNAC.from_weights(hcfg, W)
    for B, groups in ((2, 8), (1, 4), (3, 12)):                        # 8 latent frames per coarse frame: 2, 1 and 3 windows of 32
        codes = osnac.synthetic_codes(ocfg, B, groups, seed=7)
        noise = osnac.synthetic_noise(ocfg, B, groups, seed=8)
        ref, inter = orc.decoder(orc.from_codes(codes), noise, return_intermediates=True)
        got = dev.decode(codes, noise)
        blk0 = dev.debug_tap("block0", B)                              # the first block sits right behind the attention layer
        assert blk0.shape == inter["block0"].shape and np.abs(blk0 - inter["block0"]).max() <= 2e-4 * np.abs(inter["block0"]).max()
        assert got.shape == ref.shape and rms(got, ref) < 1e-4, (B, groups, rms(got, ref))
    # a latent length that is not a whole number of windows fails like the reference's reshape, loudly
    with pytest.raises(mas.AudioGenerationError):
        dev.decode(osnac.synthetic_codes(ocfg, 1, 3), None)
    rng = np.random.default_rng(5)
    for B, n in ((2, 3000), (1, 100)):
        audio = (0.3 * rng.standard_normal((B, n))).astype(np.float32)
        assert dev.padded_length(n) == orc.preprocess(audio[:, None]).shape[-1]
        codes, z = dev.encode(audio, return_latent=True)
        zr = orc.encoder(orc.preprocess(audio[:, None]))
        assert z.shape == zr.shape and np.abs(z - zr).max() <= 2e-4 * np.abs(zr).max()
        rcodes, dist = orc.encode(audio[:, None], return_details=True)
        assert len(codes) == 4
        for g, r, d in zip(codes, rcodes, dist):
            ds = np.sort(d, -1)
            sure = (ds[..., 1] - ds[..., 0]) > 1e-4
            assert g.shape == r.shape and np.array_equal(g[sure], r[sure])
