"""-m gpu: Descript DAC ((f)1) against oracle/dac.py: decoder block taps and waveform, the reference's length pins, and the encode side
(encoder latent, residual-VQ codes, encode_audio / decode_audio round trip)."""
import numpy as np
import pytest

import mlx_audio_swift_amd as mas
from oracle import dac as od

pytestmark = pytest.mark.gpu


def _pair(ocfg):
    W = od.make_synthetic_weights(ocfg)
    # checkpoint naming (before sanitize): ".layers." segments and in_proj/out_proj (DescriptDAC.swift:274-286)
    stored = {k.replace(".outProj.", ".out_proj.").replace(".inProj.", ".in_proj.").replace("decoder.model.", "decoder.model.layers."): v
              for k, v in W.items()}
    hcfg = mas.DescriptDACConfig(**{k: getattr(ocfg, k) for k in mas.DescriptDACConfig.__dataclass_fields__})
    return od.DacOracle(ocfg, W), mas.DescriptDAC.from_weights(hcfg, stored)


@pytest.mark.parametrize("ocfg", [od.TINY, od.DacConfig(encoder_dim=4, encoder_rates=(2, 2), latent_dim=40, decoder_dim=128,
                                                         decoder_rates=(8, 5), n_codebooks=2, codebook_size=64)], ids=["r324", "r85"])
def test_decode_matches_oracle(ocfg):
    orc, dev = _pair(ocfg)
    rng = np.random.default_rng(0)
    for B, T in ((2, 6), (1, 1), (1, 37)):
        codes = rng.integers(0, ocfg.codebook_size, (B, ocfg.n_codebooks, T)).astype(np.int32)
        for bi in range(len(ocfg.decoder_rates)):
            ref = orc.decode_from_codes(codes, stop_after=f"block{bi}")
            got = dev.debug_tap(codes, bi)
            assert got.shape == ref.shape and np.abs(got - ref).max() <= 3e-4 * np.abs(ref).max(), (bi, B, T)
        ref = orc.decode_from_codes(codes)
        got = dev.decode_from_codes(codes)
        assert got.shape == ref.shape == (B, od.num_samples(ocfg, T)) and np.abs(got - ref).max() <= 3e-4


def test_length_pins_of_the_reference_tests():
    # Tests/MLXAudioCodecsTests.swift:1127-1194 (decoded.shape[1])
    for rates, frames, want in (((8, 5, 4, 2), 250, 80_043), ((8, 5, 4, 2), 375, 120_043), ((8, 8, 4, 2), 430, 220_235)):
        cfg = mas.DescriptDACConfig(decoder_rates=rates, decoder_dim=16, latent_dim=8, n_codebooks=1, codebook_size=4)
        m = mas.DescriptDAC(cfg)
        assert m.num_samples(frames) == want


ENC_CFGS = [od.TINY, od.DacConfig(encoder_dim=16, encoder_rates=(2, 4, 5), latent_dim=40, decoder_dim=128, decoder_rates=(8, 5),
                                  n_codebooks=4, codebook_size=64)]


@pytest.mark.parametrize("ocfg", ENC_CFGS, ids=["r22", "r245"])
def test_encode_matches_oracle(ocfg):
    orc, dev = _pair(ocfg)
    rng = np.random.default_rng(5)
    hop = int(np.prod(ocfg.encoder_rates))
    for B, n in ((2, 9 * hop), (1, hop), (3, 7 * hop - 3)):
        audio = (0.3 * rng.standard_normal((B, n))).astype(np.float32)
        ref_codes, ref_z, margins = orc.encode(audio, return_latent=True, return_margins=True)
        codes, z = dev.encode(audio, return_latent=True)
        assert z.shape == ref_z.shape and np.abs(z - ref_z).max() <= 3e-4 * max(1.0, np.abs(ref_z).max()), (B, n)
        assert codes.shape == ref_codes.shape == (B, ocfg.n_codebooks, -(-n // hop)) and codes.dtype == np.int32
        # integer outputs: equal wherever the oracle's own top-2 distance margin is above float rounding (a near-tie in an early
        # codebook changes the residual for the later ones, so compare up to the first near-tie per frame)
        ok = np.cumprod(margins > 2e-4, axis=1).astype(bool)
        assert ok.mean() > 0.9 and (codes[ok] == ref_codes[ok]).all(), (B, n, ok.mean())
        nq = ocfg.n_codebooks - 1
        np.testing.assert_array_equal(dev.encode(audio, n_quantizers=nq), codes[:, :nq])


def test_encode_audio_round_trip_and_errors():
    ocfg = ENC_CFGS[1]
    orc, dev = _pair(ocfg)
    audio = (0.3 * np.random.default_rng(6).standard_normal((2, 203))).astype(np.float32)
    enc = dev.encode_audio(audio)
    assert enc["original_length"] == 203 and enc["codes"].shape == (2, ocfg.n_codebooks, 6)                 # 203 -> 240 = 6 * 40
    wav = dev.decode_audio(enc)
    assert wav.shape == (2, 203)
    np.testing.assert_allclose(wav, orc.decode_from_codes(enc["codes"])[:, :203], atol=3e-4)
    with pytest.raises(mas.AudioGenerationError):
        dev.preprocess(audio, sample_rate=ocfg.sample_rate + 1)
    # a decoder-only checkpoint refuses to encode (no CPU fallback, no silent zeros)
    W = {k: v for k, v in od.make_synthetic_weights(ocfg).items() if not k.startswith("encoder.") and ".inProj." not in k}
    hcfg = mas.DescriptDACConfig(**{k: getattr(ocfg, k) for k in mas.DescriptDACConfig.__dataclass_fields__})
    dec_only = mas.DescriptDAC.from_weights(hcfg, W)
    with pytest.raises(mas.AudioGenerationError):
        dec_only.encode(audio)
