"""Generates the committed golden fixtures from the CPU oracle (run in the build container):
    python tests/golden/make_golden.py
The reference itself (Swift + MLX) cannot run here and pins no numbers on this path, so these are
ORACLE outputs on seeded synthetic inputs ("parity unpinned", see oracle/__init__.py); they pin the
oracle against accidental edits and give the GPU tests a fixed target that needs no oracle run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orpheus_codes as oc  # noqa: E402
from oracle import snac  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def reference_fixture():
    """Pins that do NOT come from this repo's oracle: the reference's own speech fixture Tests/media/intention.wav (used by its codec /
    STT / smoke tests, e.g. Tests/MLXAudioSmokeTests.swift:78-110) run through an INDEPENDENT implementation of the Whisper front
    end (HF transformers' WhisperFeatureExtractor).  Needs /root/reference (build container only); the outputs are committed."""
    import shutil
    import wave
    from math import gcd

    from scipy.signal import resample_poly
    from transformers import WhisperFeatureExtractor
    src = "/root/reference/Tests/media/intention.wav"
    dst = os.path.join(OUT, "intention.wav")
    shutil.copyfile(src, dst)
    os.chmod(dst, 0o644)
    w = wave.open(dst)
    sr, n = w.getframerate(), w.getnframes()
    pcm = np.frombuffer(w.readframes(n), dtype="<i2").astype(np.float32) / 32768.0
    g = gcd(sr, 16000)
    pcm16 = resample_poly(pcm.astype(np.float64), 16000 // g, sr // g).astype(np.float32)        # 24 kHz -> 16 kHz
    out = {"sr": np.int32(sr), "pcm16k": pcm16}
    for n_mels in (80, 128):
        fe = WhisperFeatureExtractor(feature_size=n_mels)
        m = fe(pcm16, sampling_rate=16000, return_tensors="np")["input_features"][0]            # [n_mels, 3000]
        out[f"hf_mel{n_mels}"] = m.astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "intention_whisper_features.npz"), **out)


def main():
    if os.path.exists("/root/reference/Tests/media/intention.wav"):
        reference_fixture()
    # C1: SNAC 24 kHz, 12 groups = 1.024 s, B = 1, seeded synthetic weights / codes / noise
    cfg = snac.SnacConfig()
    W = snac.make_synthetic_weights(cfg, seed=1234)
    codes = snac.synthetic_codes(cfg, 1, 12, seed=1235)
    noise = snac.synthetic_noise(cfg, 1, 12, seed=1236)
    o = snac.SnacOracle(cfg, W)
    y = o.decode(codes, noise)
    y0 = o.decode(codes, None)
    np.savez_compressed(os.path.join(OUT, "snac_c1.npz"), pcm_noise=y[0, 0].astype(np.float32),
                        pcm_zero_noise=y0[0, 0].astype(np.float32), l0=codes[0], l1=codes[1], l2=codes[2])
    # framing: a ragged token stream and its parse / de-interleave
    rng = np.random.default_rng(99)
    rows, parsed = [], []
    for n_frames, junk in ((5, 3), (0, 0), (12, 6), (1, 0)):
        body = []
        for g in range(n_frames):
            body += [oc.AUDIO_TOKEN_OFFSET + k * 4096 + int(rng.integers(0, 4096)) for k in range(7)]
        ids = [oc.START_OF_HUMAN, 11, 22, oc.END_OF_TEXT, oc.END_OF_HUMAN, oc.START_OF_SPEECH] + body + \
              [oc.AUDIO_TOKEN_OFFSET + 5] * junk + [oc.END_OF_SPEECH]
        rows.append(np.asarray(ids, np.int32))
        parsed.append(oc.parse_output_row(ids))
    np.savez_compressed(os.path.join(OUT, "orpheus_framing.npz"),
                        **{f"ids{i}": r for i, r in enumerate(rows)}, **{f"codes{i}": p for i, p in enumerate(parsed)})
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
