"""Secondary bench (BASELINE configs[4] / SURVEY §8d C5, one GPU's share): Qwen3-TTS-0.6B-shaped synthetic model, batch 32, 100 frames
(8 s) per row, EOS out of reach.  Three measurements: the frame loop alone, generate with one whole-sequence decode, and
generateStream (streaming_interval 2.0 s = 25 frames: streaming steps on a second stream while the loop runs) with the time to the
first audio chunk.  argv[1] = batch (default 32), argv[2] = frames (default 100), argv[3] = 16 (bf16 weights, default) | 8 | 4 (every
2-D talker tensor as an MLX affine-quantised matrix, the published checkpoint's form).  --gpus N: N replicas, N x batch rows sharded
inside the library by mis_qwen3tts_group_generate, every replica streaming its own rows' chunks (configs[4] as worded)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_swift_amd as mas
from mlx_audio_swift_amd.synthetic import qwen3tts_synthetic_weights

NG = 1
if "--gpus" in sys.argv:
    k = sys.argv.index("--gpus")
    NG = int(sys.argv[k + 1])
    del sys.argv[k:k + 2]
B = (int(sys.argv[1]) if len(sys.argv) > 1 else 32) * NG
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = mas.Qwen3TTSConfiguration(codec_eos_token_id=3071)            # inside the suppressed range but exempt: never the argmax in practice
t0 = time.perf_counter()
BITS = int(sys.argv[3]) if len(sys.argv) > 3 else 16


def build(device):
    mm = mas.Qwen3TTSModel(cfg, device)
    for name, arr in qwen3tts_synthetic_weights(cfg):
        if BITS != 16 and arr.ndim == 2 and not name.startswith("decoder.") and arr.shape[1] % 64 == 0:
            from mlx_audio_swift_amd.synthetic import mlx_affine_quantize
            wq, sc, bi = mlx_affine_quantize(arr, 64, BITS)
            mm.set_quantized_tensor(name, wq, sc, bi, 64, BITS)
        else:
            mm.set_tensor(name, arr)
    mm.finalize()
    return mm


reps = [build(i) for i in range(NG)]
m = reps[0]
REPL = reps if NG > 1 else None
lib = mas._lib.lib()
native = [lib.mis_tts_native_quant_bits(lib.mis_qwen3tts_talker(m._h), r) for r in range(5)]
t_load = time.perf_counter() - t0
rng = np.random.default_rng(1238)
prompts = []
for b in range(B):
    text = rng.integers(0, 151000, 24)
    t = list(text[:3]) + [cfg.tts_pad_token_id] * 3 + [cfg.tts_bos_token_id] + [int(text[3])]
    c = [-1, -1, -1, cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id, cfg.codec_pad_id, cfg.codec_bos_id]
    prompts.append(mas.PreparedPrompt(np.asarray(t, np.int32), np.asarray(c, np.int32),
                                      np.asarray(list(text[4:]) + [cfg.tts_eos_token_id], np.int32), 0))
gp = mas.Qwen3TTSGenerateParameters(max_tokens=F, temperature=0.9, top_k=50, repetition_penalty=1.05, seed=9)
res = {}
for rep in range(2):
    t0 = time.perf_counter(); codes = m.generate_codes(prompts[: B // NG], gp); t_codes = time.perf_counter() - t0
    t0 = time.perf_counter(); pcm = m.generate_batch(prompts, gp, replicas=REPL); t_all = time.perf_counter() - t0
# generateStream: first-audio latency and total time with the decoder overlapped with the frame loop
for rep in range(2):
    t0 = time.perf_counter(); t_first = None; n_audio = 0; samples = 0
    if NG > 1:                      # group streaming: chunks arrive through the callback with global row indices
        def on_audio(row, a):
            global t_first, n_audio, samples
            if t_first is None:
                t_first = time.perf_counter() - t0
            n_audio += 1; samples += len(a)
        m.generate_batch(prompts, gp, streaming_interval=2.0, on_audio=on_audio, replicas=REPL)
    else:
        for ev in m.generate_stream_batch(prompts, gp, streaming_interval=2.0):
            if isinstance(ev, mas.AudioEvent):
                if t_first is None:
                    t_first = time.perf_counter() - t0
                n_audio += 1; samples += len(ev.audio)
    t_stream = time.perf_counter() - t0
audio_s = sum(len(p) for p in pcm) / cfg.sample_rate
print(json.dumps({"workload": f"Qwen3-TTS-0.6B-shaped, weights {BITS} bit (native roles {native}), batch {B} over {NG} GPU(s), {F} frames/row, 16 code groups", "n_gpus": NG,
                  "stream_total_ms": t_stream * 1e3, "stream_first_audio_ms": (t_first or 0) * 1e3, "stream_audio_events": n_audio,
                  "stream_audio_s_per_s": samples / cfg.sample_rate / t_stream,
                  "load_s": t_load, "frames": [len(c) for c in codes][:4], "codes_ms": t_codes * 1e3,
                  "ms_per_frame": t_codes * 1e3 / F, "generate_ms": t_all * 1e3, "decode_ms": (t_all - t_codes) * 1e3,
                  "audio_s_per_s": audio_s / t_all, "audio_s": audio_s}))
