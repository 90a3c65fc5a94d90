"""Probe: do two Qwen3-TTS frame loops on two streams overlap?  Two handles (own weights) run generate_codes from two host threads,
16 rows each, against one handle with 32 rows.  The frame loop is launch / latency bound (~815 small kernels per frame), so two
independent half batches should co-execute if the queues really run concurrently."""
import json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas
from mlx_audio_swift_amd.synthetic import qwen3tts_synthetic_weights

F = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = mas.Qwen3TTSConfiguration(codec_eos_token_id=3071)


def build():
    m = mas.Qwen3TTSModel(cfg)
    for name, arr in qwen3tts_synthetic_weights(cfg):
        m.set_tensor(name, arr)
    m.finalize()
    return m


def prompts(B, seed):
    rng = np.random.default_rng(seed)
    out = []
    for b in range(B):
        text = rng.integers(0, 151000, 24)
        t = list(text[:3]) + [cfg.tts_pad_token_id] * 3 + [cfg.tts_bos_token_id] + [int(text[3])]
        c = [-1, -1, -1, cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id, cfg.codec_pad_id, cfg.codec_bos_id]
        out.append(mas.PreparedPrompt(np.asarray(t, np.int32), np.asarray(c, np.int32), np.asarray(list(text[4:]) + [cfg.tts_eos_token_id], np.int32), 0))
    return out


gp = mas.Qwen3TTSGenerateParameters(max_tokens=F, temperature=0.9, top_k=50, repetition_penalty=1.05, seed=9)
a, b = build(), build()
res = {}
for B in (32, 16, 8):
    p = prompts(B, 1)
    a.generate_codes(p, gp)
    t0 = time.perf_counter(); a.generate_codes(p, gp); res[f"one_handle_b{B}_ms_per_frame"] = (time.perf_counter() - t0) * 1e3 / F
for B in (16, 8):
    pa, pb = prompts(B, 1), prompts(B, 2)
    for rep in range(2):
        th = [threading.Thread(target=m.generate_codes, args=(p, gp)) for m, p in ((a, pa), (b, pb))]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        res[f"two_handles_2x{B}_ms_per_frame"] = (time.perf_counter() - t0) * 1e3 / F
print(json.dumps(res))
