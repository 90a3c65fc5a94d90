"""Oracle vs the committed golden fixtures (tests/golden/make_golden.py)."""
import os

import numpy as np

from oracle import orpheus_codes as oc
from oracle import snac

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_framing_golden():
    z = np.load(os.path.join(G, "orpheus_framing.npz"))
    for i in range(4):
        got = oc.parse_output_row(z[f"ids{i}"])
        assert np.array_equal(got, z[f"codes{i}"])
        if len(got):
            l0, l1, l2 = oc.deinterleave(got)
            assert l0.min() >= 0 and max(l0.max(), l1.max(), l2.max()) < 4096
            assert np.array_equal(oc.interleave(l0, l1, l2), got)
    assert len(z["codes0"]) == 35 and len(z["codes1"]) == 0 and len(z["codes2"]) == 84 + 0 and len(z["codes3"]) == 7


def test_snac_c1_golden():
    z = np.load(os.path.join(G, "snac_c1.npz"))
    cfg = snac.SnacConfig()
    W = snac.make_synthetic_weights(cfg, seed=1234)
    codes = snac.synthetic_codes(cfg, 1, 12, seed=1235)
    assert np.array_equal(codes[0], z["l0"]) and np.array_equal(codes[2], z["l2"])
    y0 = snac.SnacOracle(cfg, W).decode(codes, None)[0, 0]
    assert y0.shape == (24576,)
    assert float(np.sqrt(np.mean((y0 - z["pcm_zero_noise"]) ** 2))) < 1e-6


def test_product_synthetic_generator_equals_oracle_generator():
    # bench.py / smoke() draw SNAC weights from the package's generator; the oracle has its own copy of
    # the same documented formula.  They must agree bit for bit (same weights on both sides of a parity check).
    import mlx_audio_swift_amd as mas
    from mlx_audio_swift_amd.synthetic import snac_synthetic_weights
    for cfgd in (snac.TINY, {}):
        ocfg = snac.SnacConfig(**cfgd)
        a = snac.make_synthetic_weights(ocfg, seed=1234)
        b = snac_synthetic_weights(mas.SNACConfig(**{k: getattr(ocfg, k) for k in mas.SNACConfig.__dataclass_fields__}), seed=1234)
        assert a.keys() == b.keys()
        for k in a:
            assert np.array_equal(a[k], b[k]), k


def _intention():
    z = np.load(os.path.join(G, "intention_whisper_features.npz"))
    return z


def test_reference_fixture_intention_wav_pins_the_mel_oracle():
    """Tests/media/intention.wav is the reference's own speech fixture (Tests/MLXAudioSmokeTests.swift:78-110, MLXAudioSTTTests.swift:61).
    The committed features come from HF transformers' WhisperFeatureExtractor - an implementation independent of oracle/mel.py -
    so this pins the oracle on real speech, not on this repo's own outputs.  Tolerance 2e-4 in the normalised log domain
    (HF computes the STFT in float64 numpy, the oracle in float32)."""
    import wave
    from math import gcd

    from scipy.signal import resample_poly

    from oracle import mel
    z = _intention()
    w = wave.open(os.path.join(G, "intention.wav"))
    assert (w.getframerate(), w.getnchannels(), w.getnframes()) == (24000, 1, 36480) and int(z["sr"]) == 24000
    pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
    g = gcd(24000, 16000)
    pcm16 = resample_poly(pcm.astype(np.float64), 16000 // g, 24000 // g).astype(np.float32)
    assert pcm16.shape == z["pcm16k"].shape == (24320,) and np.abs(pcm16 - z["pcm16k"]).max() < 1e-6
    for n_mels in (80, 128):
        got = mel.encoder_features(z["pcm16k"], n_mels)[0].T              # [n_mels, 3000]
        ref = z[f"hf_mel{n_mels}"]
        assert got.shape == ref.shape == (n_mels, 3000)
        d = np.abs(got - ref)
        assert d.max() < 2e-4, (n_mels, d.max())
        assert ref[:, :152].std() > 0.2                                   # real speech: the compared region is not a constant floor
