"""Debug: Qwen3 speech-tokenizer decoder stage taps, split-bf16 path against the exact-f32 kernels (same process, env toggled)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_qwen3tts as tq
cfg, dev, _, odec = tq._pair()
d = cfg.decoder
rng = np.random.default_rng(1)
for B, T in ((2, 7), (1, 40)):
    codes = rng.integers(0, d.codebook_size, (B, d.num_quantizers, T)).astype(np.int32)
    stages = [(1, "quantizer"), (2, "transformer"), (3, "upsample")] + [(4 + i, f"block{i}") for i in range(len(d.upsample_rates))]
    for sid, name in stages:
        os.environ["MIS_CODEC_EXACT_F32"] = "1"
        ex = dev.decoder_tap(codes, sid)
        os.environ["MIS_CODEC_EXACT_F32"] = "0"
        got = dev.decoder_tap(codes, sid)
        ref = odec.decode(codes, stop_after=name)
        print(B, T, name, got.shape, "exact-vs-oracle", float(np.abs(ex - ref).max() / np.abs(ref).max()),
              "bf3-vs-exact", float(np.abs(got - ex).max() / np.abs(ex).max()), flush=True)
        if np.abs(got - ex).max() > 1e-3 * np.abs(ex).max():
            bad = np.argwhere(np.abs(got - ex) > 1e-3 * np.abs(ex).max())
            print("   first bad", bad[:5].tolist(), "count", len(bad), "of", got.size, "cols", sorted(set(bad[:, 2].tolist()))[:20],
                  "rows", sorted(set(bad[:, 1].tolist()))[:20])
print("dims", d)
