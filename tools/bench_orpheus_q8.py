"""The headline workload on an 8-bit checkpoint, stand-alone (bench.py runs the same thing in its `secondary` block): Orpheus-3B dimensions, MLX
affine codes (group 64) streamed natively, batch 32, 32-token prompts, 672 new tokens, full-vocabulary sampler, SNAC 24 kHz decode.
Usage: python tools/bench_orpheus_q8.py [bits=8]   (wrap in rocprofv3 --kernel-trace --stats for the per-kernel table)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import mlx_audio_swift_amd as mas  # noqa: E402
from mlx_audio_swift_amd.synthetic import snac_synthetic_weights  # noqa: E402
import bench  # noqa: E402

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 8
snac_cfg = mas.SNACConfig()
codec = mas.SNAC.from_weights(snac_cfg, snac_synthetic_weights(snac_cfg, seed=1234))
lm_cfg = mas.LlamaTTSConfiguration(rope_theta=500000.0, rope_scaling={"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                                                       "original_max_position_embeddings": 8192, "rope_type": "llama3"})
lm = mas.LlamaTTSModel.synthetic(lm_cfg, codec=codec, seed=4321, quant_bits=bits)
prompts = bench.make_prompts(bench.ROWS_PER_GPU, 0)
flat, lens = lm._flatten(prompts)
gp = mas.GenerateParameters(max_tokens=bench.NEW_TOKENS, temperature=0.6, top_p=0.8, repetition_penalty=1.3, repetition_context_size=20, seed=2024,
                            frame_constrained=2)
gpc = gp.to_c()
n_samples = codec.num_samples(bench.NEW_TOKENS // 7)
pcm = torch.zeros((bench.ROWS_PER_GPU, n_samples), dtype=torch.float32, device="cuda:0")
plens = (C.c_int64 * bench.ROWS_PER_GPU)()
ntok = (C.c_int32 * bench.ROWS_PER_GPU)()
L = mas._lib.lib()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    st = L.mis_tts_generate_device(lm._h, flat.ctypes.data, lens.ctypes.data, bench.ROWS_PER_GPU, C.byref(gpc), None, pcm.data_ptr(), n_samples, plens, ntok)
    if st != 0:
        raise RuntimeError(mas._lib.last_error())
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0) if rep else best
t = lm.last_timing()
print(json.dumps({"workload": f"Orpheus-3B {bits}-bit checkpoint (native roles {lm.native_quant_bits}), batch 32, 672 new tokens", "audio_s_per_s": float(sum(plens)) / 24000.0 / best,
                  "ms": best * 1e3, "step_ms": t["step_ms_avg"], "hbm_GB_per_step": t["hbm_bytes_per_step"] / 1e9,
                  "frac_of_8TBps": t["hbm_bytes_per_step"] / max(t["step_ms_avg"], 1e-9) / 1e6 / 8000.0}))
