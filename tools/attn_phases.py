"""Where the decode-attention kernel spends its time: phase timestamps (s_memtime) of block (0, 0), wave 0, written by the
diagnostics build of the library (`make -C mlx-audio-swift_amd/csrc timing`).  Orpheus-3B shape, batch 32, synthetic weights.
Usage: python tools/attn_phases.py [new_tokens ...]   (context at the last step = 32 + new_tokens - 1)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MIS_LIB_PATH"] = os.path.join(ROOT, "mlx-audio-swift_amd", "libmi_speech_timing.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import mlx_audio_swift_amd as mas  # noqa: E402

from mlx_audio_swift_amd.synthetic import snac_synthetic_weights  # noqa: E402

snac_cfg = mas.SNACConfig()
codec = mas.SNAC.from_weights(snac_cfg, snac_synthetic_weights(snac_cfg, seed=1234))
cfg = mas.LlamaTTSConfiguration(rope_theta=500000.0, rope_scaling={"factor": 32.0, "rope_type": "llama3"})
lm = mas.LlamaTTSModel.synthetic(cfg, codec=codec, seed=1)
lib = mas._lib.lib()
lib.mis_debug_attn_timing.restype = C.c_int
names = ["state+slabs landed", "KV prefetch issued", "slab sum in LDS", "RoPE, q/k in LDS", "tiles processed", "partials in LDS",
         "combined + stored"]
rng = np.random.default_rng(0)
prompts = [np.concatenate([rng.integers(0, 100000, size=31), [128257]]).astype(np.int32) for _ in range(32)]   # ends with start-of-speech
for new_tokens in [int(a) for a in sys.argv[1:]] or [7, 399]:
    params = mas.GenerateParameters(max_tokens=new_tokens, temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=3,
                                    frame_constrained=True)
    assert lib.mis_debug_attn_timing_init() == 0
    lm.generate_batch(prompts, params)
    buf = (C.c_ulonglong * (4096 * 16))()
    n = lib.mis_debug_attn_timing(buf, 4096)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16)[:n].astype(np.int64)
    last = a[-28:]                                  # the decode graph's 28 launches, as of the last replay
    d = np.diff(last[:, :8], axis=1).astype(np.float64)
    tot = (last[:, 7] - last[:, 0]).mean()
    print(f"new_tokens={new_tokens}: s_memtime ticks start->end, mean over the layers: {tot:.0f}  (about 2.0 ticks/ns on MI355X - calibrate "
          f"against the kernel duration in a rocprofv3 trace; roughly {tot / 2000:.1f} us)")
    for i, nm in enumerate(names):
        print(f"   {nm:24s} {d[:, i].mean():8.1f} ticks")
