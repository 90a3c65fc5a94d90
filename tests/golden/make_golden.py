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


def main():
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
