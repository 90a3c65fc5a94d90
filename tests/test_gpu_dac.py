"""-m gpu: Descript DAC decoder ((f)1) against oracle/dac.py: block taps and waveform, plus the reference's length pins."""
import numpy as np
import pytest

import mlx_audio_swift_amd as mas
from oracle import dac as od

pytestmark = pytest.mark.gpu


def _pair(ocfg):
    W = od.make_synthetic_weights(ocfg)
    # checkpoint naming (before sanitize): ".layers." segments and in_proj/out_proj (DescriptDAC.swift:274-286)
    stored = {k.replace(".outProj.", ".out_proj.").replace("decoder.model.", "decoder.model.layers."): v for k, v in W.items()}
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
