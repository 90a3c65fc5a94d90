"""Quantised weight-streaming GEMM (csrc/lm_qgemm.hip) at bench widths: achieved GB/s of algorithmic bytes (codes + scale/bias pairs)
and time per launch, next to the dense bf16 kernel.  Usage: python tools/qgemm_probe.py [orpheus|qwen3] [batch]
MIS_QGEMM_V2=0 sends the one-shot launches through the streaming kernel (the other switches of rounds 2-3 are gone), where it
applies (read once per process); MIS_PROBE_BITS="16,8,4" picks the weight formats.  Appends to gpurun_out/qgemm_probe.jsonl."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas  # noqa: E402

which_model = sys.argv[1] if len(sys.argv) > 1 else "orpheus"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
if which_model == "orpheus":      # Orpheus-3B per-layer shapes, 8 layers (rotation defeats the Infinity Cache for the dense weights)
    cfg = mas.LlamaTTSConfiguration(num_hidden_layers=8, rope_theta=500000.0, rope_scaling={"factor": 32.0, "rope_type": "llama3"},
                                    tie_word_embeddings=False)
else:                             # Qwen3-TTS-0.6B talker shapes
    cfg = mas.LlamaTTSConfiguration(hidden_size=1024, num_hidden_layers=28, intermediate_size=3072, num_attention_heads=16,
                                    num_key_value_heads=8, head_dim=128, vocab_size=3072, rms_norm_eps=1e-6, rope_theta=1e6,
                                    rope_scaling=None, tie_word_embeddings=False, qk_norm=True, rope_plain=True)
names = ["qkv", "o_proj", "gate_up", "down", "lm_head"]
rows = []
want = [int(b) for b in os.environ.get("MIS_PROBE_BITS", "16,8,4").split(",")]
for bits in [None if b == 16 else b for b in want]:
    lm = mas.LlamaTTSModel.synthetic(cfg, seed=1, quant_bits=bits)
    out = {"model": which_model, "batch": batch, "bits": bits or 16, "qgemm_u": os.environ.get("MIS_QGEMM_U", "2"),
           "one_shot": {k[10:].lower(): v for k, v in os.environ.items() if k.startswith("MIS_QGEMM_V2")},
           "lib": os.path.basename(os.environ.get("MIS_LIB_PATH", "libmi_speech.so")), "native": lm.native_quant_bits}
    for w in range(5):
        ms, by = lm.time_gemm(w, batch, iters=64)
        out[names[w]] = {"us": round(ms * 1e3, 2), "MB": round(by / 1e6, 2), "GBps": round(by / ms / 1e6, 1)}
    print(json.dumps(out), flush=True)
    rows.append(out)
    del lm
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "qgemm_probe.jsonl"), "a") as f:
    for r in rows:
        f.write(json.dumps(r) + "\n")
