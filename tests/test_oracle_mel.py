"""Pin the log-mel oracle: HF WhisperFeatureExtractor (independent implementation) + the reference's own
known answers (Tests/MLXAudioSTTTests.swift:4416-4422, Tests/MLXAudioCodecsTests.swift:117-131)."""
import numpy as np
import pytest

from oracle import mel


def _speechlike(n, seed=0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * 220 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.05 * rng.standard_normal(n)
    x[: n // 7] *= 0.01
    return x.astype(np.float32)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_matches_hf_whisper_feature_extractor(n_mels):
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor(feature_size=n_mels)
    audio = _speechlike(16000 * 7, seed=n_mels)
    ref = fe(audio, sampling_rate=16000, return_tensors="np")["input_features"][0]      # [n_mels, 3000]
    got = mel.encoder_features(audio, n_mels)[0].T
    assert got.shape == ref.shape == (n_mels, 3000)
    assert np.abs(got - ref).max() < 2e-4
    f_ref = np.asarray(fe.mel_filters, np.float32)                                         # [201, n_mels]
    f_got = mel.mel_filters(16000, 400, n_mels, 0.0, 8000.0, "slaney", "slaney")
    assert np.abs(f_got - f_ref).max() < 1e-6


def test_reference_known_answers():
    # Tests/MLXAudioSTTTests.swift:4416-4422: 5 s of zeros, 80 mels -> [1, 3000, 80]; value = -1.5
    feats = mel.encoder_features(np.zeros(16000 * 5, np.float32), 80)
    assert feats.shape == (1, 3000, 80)
    assert np.abs(feats + 1.5).max() < 1e-6          # float32(1e-10) is not exact: log10 gives -10.000001
    # Tests/MLXAudioCodecsTests.swift:117-131: Hamming spot values
    w = mel.hamming_window(400, periodic=True)
    assert w.shape == (400,) and abs(w[0] - 0.08) < 1e-6 and abs(w[200] - 1.0) < 1e-6
    ws = mel.hamming_window(5, periodic=False)
    assert abs(ws[0] - 0.08) < 1e-6 and abs(ws[2] - 1.0) < 1e-6 and abs(ws[4] - 0.08) < 1e-6


def test_reflect_pad_and_short_inputs():
    a = np.arange(5, dtype=np.float32)
    assert mel.reflect_pad(a, 2).tolist() == [2, 1, 0, 1, 2, 3, 4, 3, 2]
    assert mel.reflect_pad(a, 6).tolist() == [0, 0] + [4, 3, 2, 1] + [0, 1, 2, 3, 4] + [3, 2, 1, 0] + [0, 0]
    assert mel.reflect_pad(np.float32([7]), 3).tolist() == [0, 0, 0, 7, 0, 0, 0]
    assert mel.log_mel_spectrogram(np.zeros(0, np.float32), 80).shape == (80, 0)      # 400 zeros -> 1 frame, dropped
    assert mel.pad_or_trim(np.ones(10, np.float32), 4).tolist() == [1, 1, 1, 1]
    assert mel.pad_or_trim(np.ones(2, np.float32), 4).tolist() == [1, 1, 0, 0]


def test_generic_dsp_path_shapes():
    x = _speechlike(16000, 3)
    m = mel.compute_mel_spectrogram(x, 16000, 400, 160, 64)
    assert m.shape == (101, 64) and np.isfinite(m).all()
    assert m.max() - m.min() <= 2.0 + 1e-6          # dynamic range clamp: 8 dB-decades / 4


def test_incremental_mel_frames_do_not_depend_on_chunking():
    # overlap-save: before the running-max clamp bites, chunked processing yields the frames of one-shot processing
    from oracle.mel import IncrementalMelOracle
    rng = np.random.default_rng(5)
    audio = (0.1 * rng.standard_normal(16000)).astype(np.float32)
    one = IncrementalMelOracle(n_mels=80)
    full = np.concatenate([x for x in (one.process(audio), one.flush()) if x is not None])
    # (the reflected prefix is built from the FIRST chunk, :77-99, so that chunk must cover n_fft/2 + 1 samples for equality)
    for sizes in ((4000, 4000, 8000), (300, 7, 333, 15360), (401, 1, 15598)):
        inc = IncrementalMelOracle(n_mels=80)
        parts, pos = [], 0
        for n in sizes:
            r = inc.process(audio[pos:pos + n]); pos += n
            if r is not None:
                parts.append(r)
        r = inc.flush()
        if r is not None:
            parts.append(r)
        got = np.concatenate(parts)
        assert got.shape == full.shape == (one.total_frames, 80) and inc.total_frames == one.total_frames
        # white noise: every chunk's maximum is within 8 of the session maximum only approximately -> compare where unclamped
        floor = (one.running_max - 8.0 + 4.0) / 4.0
        ok = (full > floor + 1e-3) & (got > floor + 1e-3)
        assert ok.mean() > 0.9 and np.abs(got - full)[ok].max() < 2e-4
