#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): audio-seconds/sec, Orpheus-3B TTS generate + SNAC decode,
batch 32 per GPU, bf16, synthetic weights + synthetic prompts (no checkpoints / datasets offline).

A "step" = one full batched generate(): 32-token prompts -> prefill -> 672 decode steps with on-device
sampling (T=0.6, top-p 0.8, repetition penalty 1.3 over 20 tokens; all 156 940 vocabulary entries go through the
sampler every step - the path of a real checkpoint - with the ids outside the step's SNAC frame slot given zero mass so that
random weights emit valid frames: frame_constrained = 2) -> parseOutput ->
de-interleave -> SNAC 24 kHz decode of 96 frames per row -> PCM resident in HBM (+ one RCCL all-gather of
the PCM when N > 1, inside the library: mis_comm_all_gather_pcm).  Inputs are resident in HBM/host-pinned-free: prompts are 4 KiB.

  python bench.py --gpus 1 --steps 2 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROWS_PER_GPU = 32
PROMPT_LEN = 32
NEW_TOKENS = 672            # 96 SNAC frames = 8.192 s per row
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def make_prompts(rows, row0, seed=1237):
    rng = np.random.default_rng(seed)
    allp = rng.integers(0, 128000, (row0 + rows, PROMPT_LEN - 4)).astype(np.int32)   # keyed by global row
    out = []
    for r in range(row0, row0 + rows):
        out.append(np.concatenate([[128259], allp[r], [128009, 128260, 128257]]).astype(np.int32))
    return out


def cpu_baseline(snac_cfg_dict):
    """Oracle (CPU restatement of the reference's MLX path) on a BOUNDED sample, rank 0 / N=1 only.
    The reference is batch-1 (LlamaTTS.swift:683-688): one utterance, sequential.  Sample = decode steps of ONE
    Orpheus-3B-shaped layer + the lm_head at context 368 (x28 layers extrapolated) + SNAC decode of 12 frames (1.024 s).
    Reproducibility: a fixed thread count (a batch-1 decode step is a chain of GEMVs - with one thread per
    core of a 128-core box the fork/join cost dominates and varies 10x between runs), 4 un-timed warm-up steps (first-touch
    page faults of 2.3 GB of f32 weights, thread-pool start), 24 timed steps, MEDIAN per step, 10th-90th percentile spread
    reported."""
    import torch
    from oracle import llama as ollama
    from oracle import snac as osnac
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(16, avail))
    torch.set_num_threads(cores)
    full = ollama.ORPHEUS_3B
    one = ollama.LlamaConfig(**{**full.__dict__, "num_hidden_layers": 1})
    g = torch.Generator().manual_seed(0)
    W = {}
    d, ff, H, Hkv, D = one.hidden_size, one.intermediate_size, one.num_attention_heads, one.num_key_value_heads, 128

    def rnd(*s):
        return (torch.rand(*s, generator=g) - 0.5) * 0.05
    W["model.embed_tokens.weight"] = rnd(one.vocab_size, d)
    W["model.norm.weight"] = torch.ones(d)
    p = "model.layers.0"
    W[p + ".input_layernorm.weight"] = torch.ones(d); W[p + ".post_attention_layernorm.weight"] = torch.ones(d)
    W[p + ".self_attn.q_proj.weight"] = rnd(H * D, d); W[p + ".self_attn.k_proj.weight"] = rnd(Hkv * D, d)
    W[p + ".self_attn.v_proj.weight"] = rnd(Hkv * D, d); W[p + ".self_attn.o_proj.weight"] = rnd(d, H * D)
    W[p + ".mlp.gate_proj.weight"] = rnd(ff, d); W[p + ".mlp.up_proj.weight"] = rnd(ff, d)
    W[p + ".mlp.down_proj.weight"] = rnd(d, ff)
    orc = ollama.LlamaOracle(one, W, round="bf16")
    orc.reset(1)
    orc.forward([np.arange(368) % 1000], logit_positions=[[367]])     # context (KV cache of 368 keys)
    warm, steps = 4, 24
    t_full, t_head = [], []
    h = torch.randn(1, d)
    for i in range(warm + steps):
        t0 = time.perf_counter()
        orc.forward([[i + 1]])                                        # one decode step: 1 layer + final norm + lm_head
        t1 = time.perf_counter()
        orc.linear(h, orc.w["model.embed_tokens.weight"])             # the lm_head part alone (tied embedding as Linear)
        t2 = time.perf_counter()
        if i >= warm:
            t_full.append(t1 - t0); t_head.append(t2 - t1)
    t_full, t_head = np.asarray(t_full), np.asarray(t_head)
    med_full, med_head = float(np.median(t_full)), float(np.median(t_head))
    # Why 16 threads and not every host core: the same step timed with ALL available cores, recorded beside the
    # 16-thread figure on every run - a batch-1 decode step is a chain of ~10 GEMVs per layer, and with one thread per core of the GPU
    # box's host the fork/join of every GEMV costs more than its arithmetic.  The baseline is the FASTER of the two thread counts.
    all_cores_med = None
    if avail > cores:
        torch.set_num_threads(avail)
        ta = []
        for i in range(2 + 8):
            t0 = time.perf_counter()
            orc.forward([[100 + i]])
            if i >= 2:
                ta.append(time.perf_counter() - t0)
        all_cores_med = float(np.median(ta))
        torch.set_num_threads(cores)
        if all_cores_med < med_full:                                    # all cores faster on this host: use them (and say so)
            scale = all_cores_med / med_full
            med_full, med_head, cores = all_cores_med, med_head * scale, avail
    t_layer = max(med_full - med_head, 1e-6)
    t_token = 28 * t_layer + med_head
    spread = (float(np.percentile(t_full, 10)), float(np.percentile(t_full, 90)))
    ocfg = osnac.SnacConfig(**snac_cfg_dict)
    SW = osnac.make_synthetic_weights(ocfg, seed=1234)
    so = osnac.SnacOracle(ocfg, SW)
    codes = osnac.synthetic_codes(ocfg, 1, 12)
    so.decode(codes, None)                                            # warm-up
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        so.decode(codes, None)
        ts.append(time.perf_counter() - t0)
    t_snac = float(np.median(ts))                                     # 1.024 s of audio
    tokens_per_audio_s = 7.0 * 24000.0 / 2048.0
    cpu_s_per_audio_s = tokens_per_audio_s * t_token + t_snac / 1.024
    return {"value": 1.0 / cpu_s_per_audio_s, "unit": "audio-s/s", "cores": int(cores), "kind": "port",
            "sample": "oracle (CPU restatement, bf16-rounded fp32 torch/numpy), BATCH 1 like the reference (the GPU line is batch 32): "
                      "%d warm-up + %d timed decode steps of 1 Orpheus-3B layer + lm_head at context 368, median per step "
                      "(x28 layers extrapolated) + median of 3 SNAC decodes of 12 frames, %d threads of %d available cores; "
                      "per-token %.4f s, SNAC %.3f s per 1.024 s" % (warm, steps, cores, avail, t_token, t_snac),
            "step_1layer_s": {"median": med_full, "p10": spread[0], "p90": spread[1], "rel_spread": (spread[1] - spread[0]) / med_full},
            "lm_head_s_median": med_head, "snac_s_per_1024ms": t_snac, "host_cores_available": int(avail),
            "threads_note": "16 threads by default: a batch-1 step is a chain of GEMVs whose fork/join over every host core costs more than the "
                            "arithmetic; the same step with all %d cores is timed on every run (step_1layer_all_cores_s) and the faster count is used" % avail,
            "step_1layer_all_cores_s": all_cores_med}


def _lm_weight_elems(cfg):
    """Elements of one decoder layer's four matrices (qkv, o, gate|up, down) of a Llama/Qwen3-style LM configuration."""
    d, ff, hd = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    H, Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
    return (H + 2 * Hkv) * hd * d + d * H * hd + 2 * ff * d + d * ff


def secondary_benches(device, orpheus=None):
    """The other single-GPU BASELINE configurations, run once each AFTER the timed region (not part of `value`): one GPU's share of
    configs[3] (Whisper-large-v3, 8 x 30 s windows) and of configs[4] (Qwen3-TTS-0.6B as the 8-bit checkpoint, batch 32, streaming), and
    configs[1] (Soprano-80M, batch 1).  Metric as the reference's CLI defines it (audio duration / wall time of the generate call,
    Sources/Tools/mlx-audio-swift-tts/App.swift:168-211).  Synthetic weights of the real dimensions; each figure is the better of two
    runs after one warm-up (graph capture, allocations).  `roofline` per entry: the bound of its dominant phase, algorithmic count /
    time / peak (2500 TFLOP/s dense bf16 MFMA, 8000 GB/s HBM; MI355X_MICROARCH.md)."""
    import mlx_audio_swift_amd as mas
    from mlx_audio_swift_amd.synthetic import mlx_affine_quantize, qwen3tts_synthetic_weights
    out = {}
    rng = np.random.default_rng(0)
    # ---- configs[3]: Whisper-large-v3, 8 windows of 30 s (mel -> encoder -> 4-token prompt + 96 greedy decode steps, EOT out of reach)
    t_begin = time.perf_counter()
    wcfg = mas.WhisperConfig(vocab_size=51866, num_mel_bins=128, d_model=1280, encoder_layers=32, encoder_attention_heads=20,
                             encoder_ffn_dim=5120, decoder_layers=32, decoder_attention_heads=20, decoder_ffn_dim=5120)
    wm = mas.WhisperModel.synthetic(wcfg, device=device, seed=777)
    wins = [(0.1 * rng.standard_normal(480000)).astype(np.float32) for _ in range(8)]
    sgp = mas.STTGenerateParameters(max_tokens=96, temperature=0.0, eot_id=-1, timestamp_begin=50365)
    feats = mas.dsp.whisper_encoder_features(np.stack(wins), 128)
    t_enc = t_all = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); wm.encode(feats, want_output=False); t_enc = min(t_enc, time.perf_counter() - t0) if rep else t_enc
        t0 = time.perf_counter(); ids = wm.transcribe_windows(wins, [50258, 50259, 50360, 50364], sgp); t_all = min(t_all, time.perf_counter() - t0) if rep else t_all
    # encoder FLOPs per window: 32 layers x (4 d^2 + 2 d ffn) x 2 x 1500 positions + attention 2 x 2 x 1500^2 x d per layer + the two convolutions
    d_, f_, T_ = 1280, 5120, 1500
    enc_flops = 8 * (32 * ((4 * d_ * d_ + 2 * d_ * f_) * 2 * T_ + 4 * T_ * T_ * d_) + 2 * 3 * 128 * d_ * 3000 + 2 * 3 * d_ * d_ * T_)
    # decoder bytes per step (the phase that is 84 % of the wall): 32 layers x (self q|k|v + o, cross q + o, fc1 + fc2 = 6 d^2 + 2 d ffn) bf16 +
    # the tied output projection once + the cross-attention K/V of the 8 windows (1500 keys x d x 2 x 2 B per window and layer) + the
    # self-attention K/V read at the mean context; 4 prompt + 96 loop steps run the layers, the output projection runs in the 96 loop steps
    n_steps = 4 + int(len(ids[0]))
    dec_w = 2.0 * 32 * (6 * d_ * d_ + 2 * d_ * f_)
    dec_head = 2.0 * 51866 * d_
    dec_cross = 8 * 32 * T_ * d_ * 2 * 2.0
    dec_self = 8 * 32 * (n_steps / 2.0) * d_ * 2 * 2.0
    dec_bytes = n_steps * (dec_w + dec_cross + dec_self) + int(len(ids[0])) * dec_head
    t_dec = max(t_all - t_enc, 1e-9)                       # (mel + scatter of the cross K/V included: an upper bound of the loop's time)
    out["whisper_large_v3_8x30s"] = {
        "config": "BASELINE configs[3], one GPU's share: Whisper-large-v3 bf16, 8 x 30 s windows, mel + encoder + 4-token prompt + 96 decode steps",
        "audio_s_per_s": 240.0 / t_all, "ms": t_all * 1e3, "encode_ms": t_enc * 1e3, "decode_ms": t_dec * 1e3,
        "decode_ms_per_step": t_dec * 1e3 / n_steps, "tokens_per_window": int(len(ids[0])),
        # BOTH phases against their own bound: the decoder loop is most of the wall (HBM: weights + cross K/V per step), the encoder is
        # the MFMA-bound part
        "roofline": {"bound": "hbm", "phase": "decoder loop (4 prompt + 96 greedy steps at 8 windows: weights + cross-attention K/V streamed per step)",
                     "achieved": dec_bytes / t_dec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dec_bytes / t_dec / 1e9 / HBM_PEAK_GBS,
                     "share_of_wall": t_dec / t_all, "bytes_per_step": (dec_w + dec_cross + dec_self + dec_head)},
        "roofline_encoder": {"bound": "mfma", "phase": "encoder (32 layers, 8 x 1500 positions)", "achieved": enc_flops / t_enc / 1e12, "peak": 2500.0,
                             "unit": "TFLOP/s", "frac": enc_flops / t_enc / 1e12 / 2500.0, "share_of_wall": t_enc / t_all}}
    wm.close()
    del wm
    # ---- configs[1]: Soprano-80M, batch 1, 24-token prompt, 64 new tokens ([STOP] out of reach) -> Vocos / ISTFT decoder
    scfg = mas.SopranoConfiguration(stop_token_id=-1)
    sm = mas.SopranoModel.synthetic(scfg, device=device, seed=4321)
    srow = [rng.integers(4, 8000, 24).astype(np.int32)]
    gps = mas.GenerateParameters(max_tokens=64, temperature=0.7, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30, seed=7, sampler_flavor=1)
    def _soprano_best():
        b = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); pcm = sm.generate_batch(srow, gps); dt = time.perf_counter() - t0
            b = min(b, dt) if rep else b
        return b, pcm
    def _soprano_stream_best():
        # generateStream, the entry point the reference's CLI times (App.swift:130-138): .token events while the launch runs, then .info / .audio
        b, first, n_tok, audio = 1e9, None, 0, None
        for rep in range(3):
            t0 = time.perf_counter(); t_first = None; n = 0
            for ev in sm.generate_stream_batch(srow, gps):
                if isinstance(ev, mas.TokenEvent):
                    n += 1
                    if t_first is None:
                        t_first = time.perf_counter() - t0
                elif isinstance(ev, mas.AudioEvent):
                    audio = ev.audio
            dt = time.perf_counter() - t0
            if rep and dt < b:
                b, first, n_tok = dt, t_first, n
        return b, first, n_tok, audio
    best, pcm_s = _soprano_best()                           # batch 1: the LM loop is ONE persistent launch (csrc/token_engine.hip)
    sop_path = sm.lm_path                                   # 1 = the token engine ran it (mis_soprano_lm_path)
    stream_best, stream_first, stream_tokens, stream_audio = _soprano_stream_best()
    stream_path = sm.lm_path
    prev = os.environ.get("MIS_TOKEN_ENGINE")
    os.environ["MIS_TOKEN_ENGINE"] = "0"                    # the same request on the launch chain (one hipGraph replay per token), for the record
    chain_best, _ = _soprano_best()
    if prev is None:
        del os.environ["MIS_TOKEN_ENGINE"]
    else:
        os.environ["MIS_TOKEN_ENGINE"] = prev
    lmc = scfg.lm_configuration()
    sop_bytes = 2.0 * (lmc.num_hidden_layers * _lm_weight_elems(lmc) + lmc.vocab_size * lmc.hidden_size)      # bf16 weights streamed per token
    out["soprano_80m_b1"] = {
        "config": "BASELINE configs[1]: Soprano-80M bf16 LM + f32 Vocos/ISTFT decoder, batch 1, 24-token prompt, 64 new tokens",
        "audio_s_per_s": len(pcm_s[0]) / scfg.sample_rate / best, "ms": best * 1e3, "ms_per_token": best * 1e3 / 64,
        "lm_loop": "token engine: the 23 leading prompt positions in ONE batched pass of the launch chain (K/V imported), then one persistent launch for "
                   "the last prompt position + 64 generated positions on the CUs of 4 XCDs, sampler inside (csrc/token_engine.hip)",
        "lm_path": sop_path,
        "generate_stream": {"audio_s_per_s": len(stream_audio) / scfg.sample_rate / stream_best, "ms": stream_best * 1e3, "first_token_ms": (stream_first or 0.0) * 1e3,
                            "token_events": stream_tokens, "lm_path": stream_path, "same_samples_as_generate": bool(np.array_equal(stream_audio, pcm_s[0])),
                            "note": "mis_soprano_generate_stream through the host mirror: .token events fired from host-visible memory while the "
                                    "persistent launch runs (Soprano.swift:801-885, App.swift:130-138)"},
        "launch_chain": {"audio_s_per_s": len(pcm_s[0]) / scfg.sample_rate / chain_best, "ms": chain_best * 1e3, "ms_per_token": chain_best * 1e3 / 64},
        "roofline": {"bound": "hbm", "phase": "decode step (LM weights streamed once per token; the whole generate call is the denominator)",
                     "achieved": sop_bytes * 64 / best / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sop_bytes * 64 / best / 1e9 / HBM_PEAK_GBS}}
    del sm
    # ---- configs[4]: Qwen3-TTS-0.6B as the 8-bit checkpoint (MLX affine, group 64, streamed natively), batch 32, 100 frames (8 s) per row,
    # generateStream with streaming_interval 2.0 s (the speech-tokenizer decoder runs on a second stream while the frame loop goes on)
    qcfg = mas.Qwen3TTSConfiguration(codec_eos_token_id=3071)
    qm = mas.Qwen3TTSModel(qcfg, device)
    for name, arr in qwen3tts_synthetic_weights(qcfg):
        if arr.ndim == 2 and not name.startswith("decoder.") and arr.shape[1] % 64 == 0:
            wq, sc, bi = mlx_affine_quantize(arr, 64, 8)
            qm.set_quantized_tensor(name, wq, sc, bi, 64, 8)
        else:
            qm.set_tensor(name, arr)
    qm.finalize()
    prompts = []
    for b in range(32):
        text = rng.integers(0, 151000, 24)
        t = list(text[:3]) + [qcfg.tts_pad_token_id] * 3 + [qcfg.tts_bos_token_id] + [int(text[3])]
        c = [-1, -1, -1, qcfg.codec_nothink_id, qcfg.codec_think_bos_id, qcfg.codec_think_eos_id, qcfg.codec_pad_id, qcfg.codec_bos_id]
        prompts.append(mas.PreparedPrompt(np.asarray(t, np.int32), np.asarray(c, np.int32), np.asarray(list(text[4:]) + [qcfg.tts_eos_token_id], np.int32), 0))
    FR = 100
    qgp = mas.Qwen3TTSGenerateParameters(max_tokens=FR, temperature=0.9, top_k=50, repetition_penalty=1.05, seed=9)
    t_codes = t_stream = 1e9
    t_first_best = None
    for rep in range(3):
        t0 = time.perf_counter(); qm.generate_codes(prompts, qgp); dt = time.perf_counter() - t0
        t_codes = min(t_codes, dt) if rep else t_codes
        t0 = time.perf_counter(); t_first = None; samples = 0
        for ev in qm.generate_stream_batch(prompts, qgp, streaming_interval=2.0):
            if isinstance(ev, mas.AudioEvent):
                if t_first is None:
                    t_first = time.perf_counter() - t0
                samples += len(ev.audio)
        dt = time.perf_counter() - t0
        if rep and dt < t_stream:
            t_stream, t_first_best = dt, t_first
    # weight bytes of one frame at 8 bit (1 byte per code + 4 bytes of bf16 scale and bias per 64 codes): the talker's 28 layers + its
    # codec head once, the code predictor's 5 layers + one of its 15 heads for each of the 15 remaining code groups
    tk, pk = qcfg.talker, qcfg.predictor
    q8 = 1.0 + 4.0 / 64.0
    frame_bytes = q8 * (tk.num_hidden_layers * _lm_weight_elems(tk) + tk.vocab_size * tk.hidden_size +
                        (qcfg.num_code_groups - 1) * (pk.num_hidden_layers * _lm_weight_elems(pk) + pk.vocab_size * pk.hidden_size))
    out["qwen3tts_0.6b_8bit_b32_stream"] = {
        "config": "BASELINE configs[4], one GPU's share: Qwen3-TTS-0.6B 8-bit checkpoint (native code streaming), batch 32, 100 frames = 8 s per row, "
                  "generateStream with a 2.0 s streaming interval",
        "audio_s_per_s": samples / qcfg.sample_rate / t_stream, "ms": t_stream * 1e3, "first_audio_ms": (t_first_best or 0.0) * 1e3,
        "ms_per_frame": t_codes * 1e3 / FR,
        "roofline": {"bound": "hbm", "phase": "frame loop (talker step + 15 code-predictor steps per frame)", "achieved": frame_bytes * FR / t_codes / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frame_bytes * FR / t_codes / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": frame_bytes}}
    del qm
    # ---- the headline workload on an 8-bit checkpoint (MLX affine quantisation, group 64; every Linear streamed as codes and dequantised
    # in registers, csrc/lm_qgemm.hip): the same 32 prompts, 672 new tokens, full-vocabulary sampler, SNAC decode
    if orpheus is not None:
        import ctypes as C
        lm_cfg, codec, flat, lens, gpc, pcm, n_samples, plens, ntok = orpheus
        lq = mas.LlamaTTSModel.synthetic(lm_cfg, codec=codec, device=device, seed=4321, quant_bits=8)
        Lb = mas._lib.lib()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            st = Lb.mis_tts_generate_device(lq._h, flat.ctypes.data, lens.ctypes.data, ROWS_PER_GPU, C.byref(gpc), None, pcm.data_ptr(), n_samples, plens, ntok)
            if st != 0:
                raise RuntimeError(mas._lib.last_error())
            import torch
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = min(best, dt) if rep else best
        tq = lq.last_timing()
        audio = float(sum(plens)) / 24000.0
        gbps = tq["hbm_bytes_per_step"] / max(tq["step_ms_avg"], 1e-9) / 1e6
        out["orpheus_3b_8bit_b32"] = {
            "config": "the headline workload (Orpheus-3B + SNAC 24 kHz, batch 32, 672 new tokens) on an 8-bit checkpoint: MLX affine codes, group 64, "
                      "streamed natively (native roles %s)" % lq.native_quant_bits,
            "audio_s_per_s": audio / best, "ms": best * 1e3, "step_ms": tq["step_ms_avg"],
            "roofline": {"bound": "hbm", "phase": "decode step (codes + scale/bias pairs + KV cache)", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbps / HBM_PEAK_GBS, "algorithmic_bytes_per_step": tq["hbm_bytes_per_step"]}}
        del lq
    out["_wall_s"] = time.perf_counter() - t_begin
    return out


def kernel_source_sha1():
    """Identity of the decode-step kernels a measurement belongs to (sha1 over the step chain's sources)."""
    import hashlib
    h = hashlib.sha1()
    for f in ("lm_kernels.hip", "lm_kernels.h", "lm_sampler.hip", "lm_engine.hip", "common.h"):
        with open(os.path.join(ROOT, "mlx-audio-swift_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (one process per GPU) the way the driver would
    (torch.distributed.run, rendezvous on 127.0.0.1) and relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup",
           str(args.warmup)] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations (Whisper / Soprano / Qwen3-TTS) after the timed region")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_spawn(args))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)      # nccl == RCCL on ROCm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    device = local_rank if world > 1 else 0
    torch.cuda.set_device(device)

    import mlx_audio_swift_amd as mas
    from mlx_audio_swift_amd.sharding import Communicator
    from mlx_audio_swift_amd.synthetic import snac_synthetic_weights

    snac_cfg = mas.SNACConfig()                                            # snac_24khz dims
    codec = mas.SNAC.from_weights(snac_cfg, snac_synthetic_weights(snac_cfg, seed=1234), device=device)
    lm_cfg = mas.LlamaTTSConfiguration(rope_theta=500000.0, rope_scaling={"factor": 32.0, "low_freq_factor": 1.0,
                                       "high_freq_factor": 4.0, "original_max_position_embeddings": 8192,
                                       "rope_type": "llama3"})            # Orpheus-3B = Llama-3.2-3B dims, vocab 156940
    lm = mas.LlamaTTSModel.synthetic(lm_cfg, codec=codec, device=device, seed=4321)

    n_rows = ROWS_PER_GPU * world
    row0 = rank * ROWS_PER_GPU
    prompts = make_prompts(ROWS_PER_GPU, row0)
    flat, lens = lm._flatten(prompts)
    gp = mas.GenerateParameters(max_tokens=NEW_TOKENS, temperature=0.6, top_p=0.8, repetition_penalty=1.3,
                                repetition_context_size=20, seed=2024, frame_constrained=2, row_offset=row0)
    gpc = gp.to_c()
    import ctypes as C
    n_samples = codec.num_samples(NEW_TOKENS // 7)
    pcm = torch.zeros((ROWS_PER_GPU, n_samples), dtype=torch.float32, device=f"cuda:{device}")
    plens = (C.c_int64 * ROWS_PER_GPU)()
    ntok = (C.c_int32 * ROWS_PER_GPU)()
    L = mas._lib.lib()

    # N > 1: the one exchange of the path - an all-gather of the decoded PCM - runs INSIDE the library (mis_comm_*: RCCL
    # ncclAllGather over xGMI on the library's stream); torch.distributed only ships the 128-byte communicator id (bootstrap)
    # and provides the barrier / max-over-ranks of the timing contract
    comm, pcm_all, gather_ms = None, None, []
    if world > 1:
        def bcast(raw):
            obj = [raw]
            dist.broadcast_object_list(obj, src=0)
            return obj[0]
        comm = Communicator(device, rank, world, bcast)
        pcm_all = torch.zeros((n_rows, n_samples), dtype=torch.float32, device=f"cuda:{device}")

    def step():
        st = L.mis_tts_generate_device(lm._h, flat.ctypes.data, lens.ctypes.data, ROWS_PER_GPU, C.byref(gpc), None,
                                       pcm.data_ptr(), n_samples, plens, ntok)
        if st != 0:
            raise RuntimeError(mas._lib.last_error())
        if comm is None:
            return pcm, np.asarray(list(plens), np.int64)
        lens_all, ms = comm.all_gather_pcm(pcm.data_ptr(), list(plens), ROWS_PER_GPU, n_samples, pcm_all.data_ptr())
        gather_ms.append(ms)
        return pcm_all, lens_all

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        allp, alll = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=pcm.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    audio_s = float(alll.sum()) / 24000.0 * args.steps
    value = audio_s / elapsed
    timing = lm.last_timing()
    # The timed region above sends the WHOLE 156 940-id vocabulary through the sampler every step (frame_constrained = 2: every id
    # visited by k_samp_cluster, the ones outside the step's frame slot get zero mass - random weights must emit valid SNAC frames
    # for there to be audio to count).  That is the code path of a real, unconstrained checkpoint.  A checkpoint-specific shortcut
    # exists (frame_constrained = 1: only the 4096 ids of the slot are visited, k_samp_narrow); the same generate is run once more
    # that way and reported beside the timed one as a secondary figure.
    narrow = None
    if rank == 0:
        gpn_ = mas.GenerateParameters(max_tokens=NEW_TOKENS, temperature=0.6, top_p=0.8, repetition_penalty=1.3,
                                      repetition_context_size=20, seed=2024, frame_constrained=True, row_offset=row0)
        gpn_c = gpn_.to_c()
        for _ in range(2):                                     # first call captures the step graph of this sampler path
            st = L.mis_tts_generate_device(lm._h, flat.ctypes.data, lens.ctypes.data, ROWS_PER_GPU, C.byref(gpn_c), None,
                                           pcm.data_ptr(), n_samples, plens, ntok)
            if st != 0:
                raise RuntimeError(mas._lib.last_error())
        tn = lm.last_timing()
        narrow = {"decode_ms": tn["decode_ms"], "step_ms": tn["step_ms_avg"], "tokens_per_row_min": int(min(ntok))}

    result = None
    if rank == 0:
        # roofline of the dominant kernel: the weight-streaming skinny GEMM (k_gemm_skinny); measured live with
        # HIP events on the library's stream, rotating over the 28 layers so the 256 MB Infinity Cache cannot
        # serve the weights.  Dominant instance by bytes/step: gate+up (42 % of the step's weight bytes).
        names = ["qkv", "o_proj", "gate_up", "down", "lm_head", "attn_decode_ctx368", "reduce_residual_rmsnorm"]
        kern = {}
        for i, n in enumerate(names):
            ms, by = lm.time_gemm(i, ROWS_PER_GPU, iters=56)
            kern[n] = {"us": ms * 1e3, "GBps": by / ms / 1e6, "bytes": by}
        dom = kern["gate_up"]
        # HBM bytes per launch of the dominant kernel from the PMC passes of tools/pmc_traffic.sh (FETCH_SIZE / WRITE_SIZE in their own
        # rocprofv3 runs, gfx950 corrections of MI355X_MICROARCH.md); tagged with the kernel source it was measured on, so a stale
        # figure is visible as such
        traffic, traffic_src = None, None
        for cand in ("r06_pmc", "r05_pmc", "r04_pmc", "r03_pmc", "r02_pmc", "r01_pmc"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", cand, "traffic.json")))
                traffic = tj["gate_up"]["hbm_bytes_per_launch"]
                traffic_src = {"file": f"profiles/{cand}/traffic.json", "measured_on_kernel_source": tj.get("kernel_source_sha1"),
                               "current_kernel_source": kernel_source_sha1(),
                               "matches_current_build": tj.get("kernel_source_sha1") == kernel_source_sha1()}
                break
            except Exception:
                continue
        step_ms = timing["step_ms_avg"]
        step_GBps = timing["hbm_bytes_per_step"] / max(step_ms, 1e-9) / 1e6
        roofline = {"bound": "hbm", "kernel": "k_gemm_skinny<MT=2,R=4,silu_mul,KSB=4,U=3> (gate+up, 100.7 MB/launch)",
                    "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": dom["bytes"],
                    "launch_us": dom["us"],
                    "kernels": {k: {"us": round(v["us"], 2), "GBps": round(v["GBps"], 1), "frac": round(v["GBps"] / HBM_PEAK_GBS, 3),
                                    "algorithmic_bytes": v["bytes"]} for k, v in kern.items()},
                    "step": {"ms": step_ms, "algorithmic_GB": timing["hbm_bytes_per_step"] / 1e9, "achieved_GBps": step_GBps,
                             "frac": step_GBps / HBM_PEAK_GBS, "frac_of_measured_copy_6290": step_GBps / 6290.0}}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline({})
        result = {
            # BASELINE.json's metric is quoted per GPU; the bench contract wants `value` = the WHOLE-JOB aggregate over n_gpus (the
            # driver derives scaling efficiency from it) - the per-GPU figure BASELINE names is `value_per_gpu`
            "metric": "audio-seconds/sec (TTS gen+codec decode), Orpheus-3B batch 32 per GPU: value = whole-job aggregate over n_gpus, "
                      "value_per_gpu = value / n_gpus",
            "value": value, "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "Orpheus-3B bf16 TTS + SNAC 24 kHz decode, batch 32 per GPU (BASELINE configs[2], the configuration the "
                                   "metric is quoted on), 32-token prompts, 672 new tokens/row = 8.192 s/row",
                       "rows_per_gpu": ROWS_PER_GPU, "global_rows": n_rows, "prompt_len": PROMPT_LEN,
                       "new_tokens": NEW_TOKENS, "parallelism": f"utterance-dp{world}", "sampler": "T0.6 top-p0.8 rep1.3"},
            "value_per_gpu": value / world,
            "phases_ms": {"prefill": timing["prefill_ms"], "decode": timing["decode_ms"], "codec": timing["codec_ms"],
                          "all_gather_rccl": (float(np.mean(gather_ms[-args.steps:])) if gather_ms else 0.0),
                          "decode_frame_slot_sampler": narrow["decode_ms"] if narrow else None},
            "sampler": {"timed_region": "full vocabulary: every one of the 156 940 ids visited by the sampler each step (k_samp_cluster, one "
                                        "launch: the path of an unconstrained checkpoint; frame_constrained = 2 gives the ids outside the "
                                        "step's frame slot zero mass so that the tokens form valid frames)",
                        "timed_step_ms": timing["step_ms_avg"],
                        "stand_in_note": "ids outside the frame slot carry zero mass and skip their histogram update: with every id live the "
                                         "sampler launch costs 36.5 us instead of 32.9 us (tools/samp_phases.py, profiles/r04/c11_samp_phases.txt) - "
                                         "about +3.6 us per step (0.17 %) for a real unconstrained checkpoint",
                        "frame_slot_step_ms": narrow["step_ms"] if narrow else None,
                        "frame_slot": "secondary: the same generate with the sampler visiting only the 4096 ids of the step's frame slot "
                                      "(k_samp_narrow, frame_constrained = 1 - a shortcut only a frame-locked checkpoint could take); "
                                      "tokens per row %s" % (narrow["tokens_per_row_min"] if narrow else None)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if world == 1 and not args.no_secondary:
            try:
                result["secondary"] = secondary_benches(device, (lm_cfg, codec, flat, lens, gpc, pcm, n_samples, plens, ntok))
            except Exception as ex:                              # the headline line must not be lost to a secondary configuration
                result["secondary"] = {"error": repr(ex)}
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
