"""-m gpu: Qwen3-TTS (a22): on-device sampleToken vs the oracle (exact), speech-tokenizer decoder vs the oracle (stage taps and
waveform), the talker + code-predictor frame loop under teacher forcing, EOS / ragged rows, and generate end to end.
Waveform tolerance: float32 chain of ~60 layers with sin/exp: max |err| <= 5e-4 * max |ref| per stage."""
import numpy as np
import pytest
import torch

import mlx_audio_swift_amd as mas
from gpu_util import lm_host_config
from oracle import qwen3tts as oq

pytestmark = pytest.mark.gpu


def _host_cfg(o: oq.Qwen3TTSConfig) -> mas.Qwen3TTSConfiguration:
    d = o.decoder
    dec = mas.Qwen3TTSDecoderConfiguration(**{k: getattr(d, k) for k in mas.Qwen3TTSDecoderConfiguration.__dataclass_fields__})
    return mas.Qwen3TTSConfiguration(
        talker=lm_host_config(o.talker), predictor=lm_host_config(o.predictor), num_code_groups=o.num_code_groups,
        text_hidden_size=o.text_hidden_size, text_vocab_size=o.text_vocab_size, codec_eos_token_id=o.codec_eos_token_id,
        codec_think_id=o.codec_think_id, codec_nothink_id=o.codec_nothink_id, codec_think_bos_id=o.codec_think_bos_id,
        codec_think_eos_id=o.codec_think_eos_id, codec_pad_id=o.codec_pad_id, codec_bos_id=o.codec_bos_id,
        tts_pad_token_id=o.tts_pad_token_id, tts_bos_token_id=o.tts_bos_token_id, tts_eos_token_id=o.tts_eos_token_id, decoder=dec)


def _pair(ocfg=oq.TINY, eos=None, quant_bits=None):
    if eos is not None:
        ocfg = oq.Qwen3TTSConfig(**{**ocfg.__dict__, "codec_eos_token_id": eos})
    W = oq.make_synthetic_weights(ocfg)
    Wd = oq.make_synthetic_decoder_weights(ocfg.decoder)
    if quant_bits:
        # a quantised checkpoint (Qwen3TTS.swift:1157-1170): every 2-D talker tensor as uint32 words + bf16 scales / biases.  Oracle
        # weights: the float32 s*q+b for the Linear layers of the two LMs (quantizedMatmul never rounds the weight), model-dtype
        # rows for everything gathered or folded (QuantizedEmbedding, text projection, predictor tables / heads)
        from oracle import mlxquant as mq
        dev = mas.Qwen3TTSModel(_host_cfg(ocfg))
        Wo = {}
        for k, v in W.items():
            v = torch.as_tensor(v)
            if v.ndim == 2 and v.shape[1] % 64 == 0:
                wq, s, bia = mq.quantize(v.float().numpy(), 64, quant_bits)
                s16, b16 = torch.from_numpy(s).bfloat16(), torch.from_numpy(bia).bfloat16()
                dev.set_quantized_tensor("talker." + k, wq, s16, b16, 64, quant_bits)
                d32 = torch.from_numpy(mq.dequantize(wq, s16.float().numpy(), b16.float().numpy(), 64, quant_bits))
                lm_linear = ("model.layers." in k and k.endswith("_proj.weight")) or k == "codec_head.weight"
                Wo[k] = d32 if lm_linear else d32.bfloat16()
            else:
                dev.set_tensor("talker." + k, v); Wo[k] = v
        for k, v in Wd.items():
            dev.set_tensor(k, v)
        dev.finalize()
        return ocfg, dev, oq.Qwen3TTSOracle(ocfg, Wo), oq.SpeechDecoderOracle(ocfg.decoder, Wd)
    allw = {("talker." + k): v for k, v in W.items()}          # checkpoint naming: sanitize strips the prefix
    allw.update(Wd)
    dev = mas.Qwen3TTSModel.from_weights(_host_cfg(ocfg), allw)
    return ocfg, dev, oq.Qwen3TTSOracle(ocfg, W), oq.SpeechDecoderOracle(ocfg.decoder, Wd)


def _prompt(cfg, rng, n_text, n_trail):
    text = list(rng.integers(0, cfg.text_vocab_size - 20, n_text))
    t = text[:3] + [cfg.tts_pad_token_id] * 2 + [cfg.tts_bos_token_id] + [text[3]]
    c = [-1, -1, -1, cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id, cfg.codec_bos_id]
    extra = text[4:]
    t = extra + t; c = [-1] * len(extra) + c                     # an "instruct" prefix of text-only positions
    trailing = list(rng.integers(0, cfg.text_vocab_size - 20, n_trail)) + [cfg.tts_eos_token_id]
    return mas.PreparedPrompt(np.asarray(t, np.int32), np.asarray(c, np.int32), np.asarray(trailing, np.int32), 0)


def test_sampler_matches_oracle_exactly():
    import ctypes as C
    from mlx_audio_swift_amd import _lib
    rng = np.random.default_rng(0)
    for V in (1280, 96, 2048):
        B = 5
        lg = torch.as_tensor((rng.standard_normal((B, V)) * 4).astype(np.float32)).bfloat16().float().numpy()
        lg[0, :8] = lg[0, 8]                                      # ties
        seen = np.zeros((B, V), np.uint8)
        gen = []
        for b in range(B):
            ids = rng.integers(0, V, 6)
            ids[0] = int(np.argmax(lg[b]))
            seen[b, ids] = 1
            gen.append(list(ids))
        sup = (V - 40, V) if V > 200 else (0, 0)
        eos = V - 10 if V > 200 else -1
        combos = [dict(temperature=0.9, top_p=1.0, top_k=50, repetition_penalty=1.05, min_p=0.0),
                  dict(temperature=0.0, top_p=1.0, top_k=0, repetition_penalty=1.3, min_p=0.0),
                  dict(temperature=1.3, top_p=0.7, top_k=0, repetition_penalty=1.0, min_p=0.0),
                  dict(temperature=0.6, top_p=0.9, top_k=20, repetition_penalty=1.5, min_p=0.05),
                  dict(temperature=1.0, top_p=1.0, top_k=0, repetition_penalty=1.0, min_p=0.2)]
        for ci, pr in enumerate(combos):
            for step in (0, 7):
                gp = mas.Qwen3TTSGenerateParameters(max_tokens=1, seed=11 + ci, **pr).to_c()
                out = np.zeros(B, np.int32)
                l32 = np.ascontiguousarray(lg)
                mas.generation.check(_lib.lib().mis_qwen3tts_sample_logits(0, l32.ctypes.data, B, V, seen.ctypes.data, C.byref(gp),
                                                                           sup[0], sup[1], eos, step, out.ctypes.data))
                for b in range(B):
                    ref = oq.sample_token(lg[b], pr["temperature"], pr["top_p"], pr["top_k"], pr["repetition_penalty"], gen[b],
                                          range(sup[0], sup[1]) if sup[1] else None and None, eos if eos >= 0 else None, pr["min_p"],
                                          11 + ci, b, step)
                    if sup[1]:
                        ref = oq.sample_token(lg[b], pr["temperature"], pr["top_p"], pr["top_k"], pr["repetition_penalty"], gen[b],
                                              [t for t in range(sup[0], sup[1]) if t != eos], eos if eos >= 0 else None, pr["min_p"],
                                              11 + ci, b, step)
                    assert out[b] == ref, (V, ci, step, b)


def test_decoder_stages_and_waveform_match_oracle():
    cfg, dev, _, odec = _pair()
    d = cfg.decoder
    rng = np.random.default_rng(1)
    for B, T in ((2, 7), (1, 1), (1, 40)):
        codes = rng.integers(0, d.codebook_size, (B, d.num_quantizers, T)).astype(np.int32)
        stages = [(1, "quantizer"), (2, "transformer"), (3, "upsample")] + [(4 + i, f"block{i}") for i in range(len(d.upsample_rates))]
        for sid, name in stages:
            ref = odec.decode(codes, stop_after=name)
            got = dev.decoder_tap(codes, sid)
            assert got.shape == ref.shape, (name, got.shape, ref.shape)
            assert np.abs(got - ref).max() <= 5e-4 * np.abs(ref).max(), (name, B, T)
        ref = odec.decode(codes)
        got = dev.decode_codes(codes)
        assert got.shape == ref.shape == (B, T * d.total_upsample) and dev.samples_per_frame == d.total_upsample
        assert np.abs(got - ref).max() <= 5e-4 * max(np.abs(ref).max(), 1e-3), (B, T)
    # causality on the device too: a prefix of the frames gives the prefix of the waveform (what streaming relies on)
    codes = rng.integers(0, d.codebook_size, (1, d.num_quantizers, 12)).astype(np.int32)
    full = dev.decode_codes(codes)
    part = dev.decode_codes(codes[:, :, :5])
    assert np.abs(part - full[:, : 5 * d.total_upsample]).max() < 1e-5


@pytest.mark.parametrize("ocfg,quant", [(oq.TINY, None), (oq.TINY_PROJ, None), (oq.TINY, 8), (oq.TINY_PROJ, 4)],
                         ids=["same-width", "mtp-projection", "8bit-checkpoint", "4bit-checkpoint-mtp"])
def test_frame_loop_greedy_under_teacher_forcing(ocfg, quant):
    cfg, dev, olm, _ = _pair(ocfg, quant_bits=quant)
    if quant:                                                                    # the talker streams every role as codes
        lib = mas._lib.lib()
        assert [lib.mis_tts_native_quant_bits(lib.mis_qwen3tts_talker(dev._h), r) for r in range(5)] == [quant] * 5
    rng = np.random.default_rng(2)
    prompts = [_prompt(cfg, rng, 9, 3), _prompt(cfg, rng, 4, 1), _prompt(cfg, rng, 6, 5)]
    gp = mas.Qwen3TTSGenerateParameters(max_tokens=6, temperature=0.0, repetition_penalty=1.05, seed=1)
    codes = dev.generate_codes(prompts, gp)
    assert [c.shape for c in codes] == [(6, cfg.num_code_groups)] * 3
    assert dev.generate_codes(prompts[1:2], gp)[0].tolist() == codes[1].tolist()          # batch row == single row
    suppress = [t for t in range(cfg.talker.vocab_size - 1024, cfg.talker.vocab_size) if t != cfg.codec_eos_token_id]
    pr = dict(temperature=0.0, top_p=1.0, top_k=0, repetition_penalty=1.05, min_p=0.0, seed=1)
    for b, p in enumerate(prompts):
        # oracle logits along the engine's own codes; the engine's choice must be the oracle's argmax within tolerance
        olm.talker.reset(1)
        x = olm.position_embeds(p.text_ids, p.codec_ids)
        trailing = olm.text_embed(p.trailing_ids)
        pad = olm.text_embed([cfg.tts_pad_token_id])[0]
        gen0 = []
        for f in range(6):
            lg = olm.talker.forward_embeds(0, x)[-1].numpy()
            hidden = olm.talker.last_hidden[-1]
            l = lg.copy(); l[suppress] = -np.inf
            pen = oq.synth.bf16_round(np.asarray([1.05], np.float32))[0]
            for t in set(gen0):
                l[t] = l[t] * pen if l[t] < 0 else l[t] / pen
            tol = 0.04 * float(np.abs(lg).max())
            c0 = int(codes[b][f, 0])
            assert c0 < cfg.talker.vocab_size - 1024 and l[c0] >= l.max() - tol, (b, f)
            _, plog = olm.predictor_codes(hidden, c0, pr, b, f, forced=codes[b][f], want_logits=True)
            for i, pl in enumerate(plog):
                ci = int(codes[b][f, i + 1])
                assert pl[ci] >= pl.max() - 0.04 * float(np.abs(pl).max()), (b, f, i)
            te = trailing[f] if f < trailing.shape[0] else pad
            x = olm.next_input(te, [int(v) for v in codes[b][f]])[None]
            gen0.append(c0)


def test_sampled_generation_eos_ragged_rows_and_end_to_end_audio():
    cfg, dev, olm, odec = _pair()
    rng = np.random.default_rng(3)
    prompts = [_prompt(cfg, rng, 7, 2), _prompt(cfg, rng, 5, 4)]
    gp = mas.Qwen3TTSGenerateParameters(max_tokens=8, temperature=0.9, top_k=50, repetition_penalty=1.05, seed=5)
    a = dev.generate_codes(prompts, gp)
    b = dev.generate_codes(prompts, gp)
    gp2 = mas.Qwen3TTSGenerateParameters(max_tokens=8, temperature=0.9, top_k=50, repetition_penalty=1.05, seed=6)
    c = dev.generate_codes(prompts, gp2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and not all(np.array_equal(x, y) for x, y in zip(a, c))
    shard = dev.generate_codes(prompts[1:], mas.Qwen3TTSGenerateParameters(max_tokens=8, temperature=0.9, top_k=50,
                                                                            repetition_penalty=1.05, seed=5, row_offset=1))
    assert np.array_equal(shard[0], a[1])                                # RNG keyed by global row
    # per-row frame caps (effectiveMaxTokens) make ragged rows
    prompts[0].target_token_count = 0
    caps = dev.generate_codes([mas.PreparedPrompt(p.text_ids, p.codec_ids, p.trailing_ids, t) for p, t in zip(prompts, (1, 0))],
                              mas.Qwen3TTSGenerateParameters(max_tokens=80, temperature=0.0))
    assert len(caps[0]) == 75 and len(caps[1]) == 80                      # max(75, 6 * tokens) vs maxTokens
    # EOS: make row 0's 4th greedy code-0 the EOS id -> that row stops after 3 frames, the other keeps going
    g = dev.generate_codes(prompts, mas.Qwen3TTSGenerateParameters(max_tokens=8, temperature=0.0))
    eos = int(g[0][3, 0])
    assert eos not in g[0][:3, 0]
    cfg2, dev2, _, _ = _pair(eos=eos)
    g2 = dev2.generate_codes(prompts, mas.Qwen3TTSGenerateParameters(max_tokens=8, temperature=0.0))
    assert np.array_equal(g2[0], g[0][:3])
    k = list(g[1][:, 0]).index(eos) if eos in g[1][:, 0] else 8
    assert np.array_equal(g2[1], g[1][:k])
    # end to end, no streaming: pcm == speech-tokenizer oracle applied to the engine's codes (rows decode together, right-padded)
    pcm, codes = dev.generate_batch(prompts, gp, return_codes=True)
    for r in range(2):
        assert np.array_equal(codes[r], a[r])
        ref = odec.decode(codes[r].T[None])[0]
        assert pcm[r].shape == ref.shape == (8 * cfg.decoder.total_upsample,)
        assert np.abs(pcm[r] - ref).max() <= 5e-4 * max(np.abs(ref).max(), 1e-3)
    # ragged rows in one padded decode: row 0 stops after 3 frames (EOS), row 1 runs on
    pr, cr = dev2.generate_batch(prompts, mas.Qwen3TTSGenerateParameters(max_tokens=8, temperature=0.0), return_codes=True)
    for r in range(2):
        assert len(pr[r]) == len(cr[r]) * cfg.decoder.total_upsample
        if len(cr[r]):
            ref = odec.decode(cr[r].T[None])[0]
            assert np.abs(pr[r] - ref).max() <= 5e-4 * max(np.abs(ref).max(), 1e-3)


def _oracle_stream(odec, codes_bqt, cuts):
    odec.reset_streaming_state()
    out, a0 = [], 0
    for b in list(cuts) + [codes_bqt.shape[-1]]:
        if b > a0:
            out.append(odec.streaming_step(codes_bqt[:, :, a0:b]))
        a0 = b
    return np.concatenate(out, -1)


def test_streaming_step_carried_state_matches_oracle_and_exact_mode_is_bitwise():
    """streamingStep (Qwen3TTSSpeechTokenizer.swift:971-1006): only the new frames are computed, conv history / transposed-conv
    overlap / K/V cache stay on the device.  exact mode: any chunking is BITWISE the whole-sequence decode.  Default mode: the
    reference's arithmetic (bias twice after a boundary), compared with the oracle's literal restatement of the step functions."""
    cfg, dev, _, odec = _pair()
    d = cfg.decoder
    up = d.total_upsample
    rng = np.random.default_rng(11)
    B, T = 2, 13
    codes = rng.integers(0, d.codebook_size, (B, d.num_quantizers, T)).astype(np.int32)
    whole = dev.decode_codes(codes)
    for cuts in ([5], [1, 2, 3, 4], [4, 5, 12], list(range(1, T))):
        bounds = [0] + list(cuts) + [T]
        dev.set_stream_exact(True)
        dev.reset_streaming_state(batch=B, max_frames=T, max_chunk_frames=max(b - a for a, b in zip(bounds, bounds[1:])))
        got = np.concatenate([dev.streaming_step(codes[:, :, a:b]) for a, b in zip(bounds, bounds[1:])], -1)
        assert np.array_equal(got, whole), ("exact", cuts, float(np.abs(got - whole).max()))
        dev.set_stream_exact(False)
        dev.reset_streaming_state(batch=B, max_frames=T, max_chunk_frames=T)
        got = np.concatenate([dev.streaming_step(codes[:, :, a:b]) for a, b in zip(bounds, bounds[1:])], -1)
        ref = _oracle_stream(odec, codes, cuts)
        err = float(np.abs(got - ref).max())
        print(f"PARITY q3 streaming cuts={cuts}: max err {err:.2e} (max |ref| {np.abs(ref).max():.3f})")
        assert err <= 5e-4 * max(np.abs(ref).max(), 1e-3), cuts
        assert np.array_equal(got[:, : cuts[0] * up], whole[:, : cuts[0] * up])          # identical up to the first boundary
        assert np.abs(got - whole).max() > 1e-5                                           # ... and the doubled bias after it
    dev.end_streaming()
    with pytest.raises(mas.AudioGenerationError):
        dev.streaming_step(codes[:, :, :1])                                               # no session open


def test_generate_stream_delivers_audio_while_the_frame_loop_runs():
    """generateStream (Qwen3TTS.swift:101-133,492-505,537-546): audio chunks come out of streaming steps that run while the
    talker keeps generating; their concatenation is the row's pcm; the first AUDIO event precedes the last TOKEN event."""
    cfg, dev, olm, odec = _pair()
    rng = np.random.default_rng(3)
    prompts = [_prompt(cfg, rng, 7, 2), _prompt(cfg, rng, 5, 4)]
    up = cfg.decoder.total_upsample
    gp = mas.Qwen3TTSGenerateParameters(max_tokens=21, temperature=0.9, top_k=50, repetition_penalty=1.05, seed=5)
    codes = dev.generate_codes(prompts, gp)                                               # (this seed: row 1 samples EOS after 18 frames)
    events = list(dev.generate_stream_batch(prompts, gp, streaming_interval=0.4))         # 5 frames per chunk
    for r in range(2):
        n = len(codes[r])
        ended_on_eos = n < 21
        toks = [e.token for e in events if isinstance(e, mas.TokenEvent) and e.row == r]
        # code 0 of every frame, in order; onToken fires for the EOS id too (Qwen3TTS.swift:484-487)
        assert toks == list(codes[r][:, 0]) + ([cfg.codec_eos_token_id] if ended_on_eos else [])
        chunks = [e.audio for e in events if isinstance(e, mas.AudioEvent) and e.row == r]
        assert [len(x) for x in chunks] == [5 * up] * (n // 5) + ([(n % 5) * up] if n % 5 else [])
        ref = _oracle_stream(odec, codes[r].T[None], list(range(5, n, 5)))[0]
        got = np.concatenate(chunks)
        assert np.abs(got - ref).max() <= 5e-4 * max(np.abs(ref).max(), 1e-3)
        infos = [e for e in events if isinstance(e, mas.InfoEvent) and e.row == r]
        assert len(infos) == 1 and infos[0].info.generation_token_count == n
    kinds = ["A" if isinstance(e, mas.AudioEvent) else "T" if isinstance(e, mas.TokenEvent) else "I" for e in events]
    first_audio, last_token = kinds.index("A"), len(kinds) - 1 - kinds[::-1].index("T")
    assert first_audio < last_token, "".join(kinds)                                       # audio while the loop was still sampling
    assert kinds.index("I") > last_token and "A" in kinds[kinds.index("I"):]              # info, then the remaining frames
    # the blocking call with a chunk callback returns the same samples as pcm; exact mode equals the whole-sequence decode bitwise
    got_chunks = {0: [], 1: []}
    pcm = dev.generate_batch(prompts, gp, streaming_interval=0.4, on_audio=lambda r, x: got_chunks[r].append(x))
    plain = dev.generate_batch(prompts, gp)
    for r in range(2):
        assert np.array_equal(np.concatenate(got_chunks[r]), pcm[r])
        assert not np.array_equal(pcm[r], plain[r])
    dev.set_stream_exact(True)
    pcm_x = dev.generate_batch(prompts, gp, streaming_interval=0.4, on_audio=lambda r, x: None)
    for r in range(2):
        assert np.array_equal(pcm_x[r], plain[r])
    # EOS mid-chunk: the row's last chunk is short, nothing follows it
    g = dev.generate_codes(prompts, mas.Qwen3TTSGenerateParameters(max_tokens=12, temperature=0.0))
    eos = int(g[0][7, 0])
    if eos not in g[0][:7, 0]:
        cfg2, dev2, _, _ = _pair(eos=eos)
        ev = list(dev2.generate_stream_batch(prompts, mas.Qwen3TTSGenerateParameters(max_tokens=12, temperature=0.0), streaming_interval=0.4))
        lens0 = [len(e.audio) for e in ev if isinstance(e, mas.AudioEvent) and e.row == 0]
        assert lens0 == [5 * up, 2 * up]
        t0 = [e.token for e in ev if isinstance(e, mas.TokenEvent) and e.row == 0]
        assert t0 == list(g[0][:7, 0]) + [eos]                                           # onToken fires for the EOS id too (:484-487)


def test_from_model_directory_quantised_checkpoint_and_pytorch_layout_tokenizer(tmp_path):
    """fromModelDirectory (Qwen3TTS.swift:1136-1275): config.json (talker_config, quantization), talker tensors under `talker.` with
    8-bit packed weights + scales / biases, speech_tokenizer/ in the PyTorch layout the published checkpoints use ([out, in, k] convs,
    [in, out, k] transposed convs, `_codebook.` statistics) -> the same model as building it tensor by tensor (codes and waveform
    bitwise).  codebook_dim 160 / decoder_dim 288 keep the reference's conv-layout heuristic (checkArrayShapeQwen3 :1445-1455: it reads
    [*, 1, <=64] and [*, <=64, 1] the wrong way round) on its valid side, as the published dimensions do."""
    import json
    import os
    from safetensors.torch import save_file
    from oracle import mlxquant as mq
    dcfg = oq.DecoderConfig(**{**oq.TINY.decoder.__dict__, "codebook_dim": 160, "decoder_dim": 288})    # 1x1 convs wider than 64 too
    ocfg = oq.Qwen3TTSConfig(**{**oq.TINY.__dict__, "decoder": dcfg})
    W = oq.make_synthetic_weights(ocfg)
    Wd = oq.make_synthetic_decoder_weights(dcfg)
    ref = mas.Qwen3TTSModel(_host_cfg(ocfg))
    tens = {}
    for k, v in W.items():
        v = torch.as_tensor(v)
        if v.ndim == 2 and v.shape[1] % 64 == 0:
            wq, s_, b_ = mq.quantize(v.float().numpy(), 64, 8)
            s16, b16 = torch.from_numpy(s_).bfloat16(), torch.from_numpy(b_).bfloat16()
            ref.set_quantized_tensor("talker." + k, wq, s16, b16, 64, 8)
            base = "talker." + k[: -len(".weight")]
            tens[base + ".weight"] = torch.from_numpy(wq.view(np.int32)).view(torch.int32)
            tens[base + ".scales"], tens[base + ".biases"] = s16, b16
        else:
            ref.set_tensor("talker." + k, v); tens["talker." + k] = v.contiguous()
    for k, v in Wd.items():
        ref.set_tensor(k, v)
    ref.finalize()
    d = tmp_path / "q3"; (d / "speech_tokenizer").mkdir(parents=True)
    lmj = lambda c: dict(hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers, intermediate_size=c.intermediate_size,
                         num_attention_heads=c.num_attention_heads, num_key_value_heads=c.num_key_value_heads, head_dim=c.head_dim,
                         vocab_size=c.vocab_size, rms_norm_eps=c.rms_norm_eps, rope_theta=c.rope_theta)
    tj = {**lmj(ocfg.talker), "code_predictor_config": lmj(ocfg.predictor), "num_code_groups": ocfg.num_code_groups,
          "text_hidden_size": ocfg.text_hidden_size, "text_vocab_size": ocfg.text_vocab_size}
    for f in ("codec_eos_token_id", "codec_think_id", "codec_nothink_id", "codec_think_bos_id", "codec_think_eos_id", "codec_pad_id", "codec_bos_id"):
        tj[f] = getattr(ocfg, f)
    (d / "config.json").write_text(json.dumps({"model_type": "qwen3_tts", "talker_config": tj, "quantization": {"group_size": 64, "bits": 8},
                                                "tts_pad_token_id": ocfg.tts_pad_token_id, "tts_bos_token_id": ocfg.tts_bos_token_id,
                                                "tts_eos_token_id": ocfg.tts_eos_token_id}))
    path = str(d / "model.safetensors")
    save_file(tens, path)
    raw = open(path, "rb").read()                                            # MLX writes the packed words as U32
    n = int.from_bytes(raw[:8], "little")
    hdr = json.loads(raw[8:8 + n])
    for k in hdr:
        if k != "__metadata__" and hdr[k]["dtype"] == "I32":
            hdr[k]["dtype"] = "U32"
    hb = json.dumps(hdr, separators=(",", ":")).encode()
    hb += b" " * ((8 - len(hb) % 8) % 8)
    open(path, "wb").write(len(hb).to_bytes(8, "little") + hb + raw[8 + n:])
    # speech tokenizer as a PyTorch-layout checkpoint
    pt = {}
    for k, v in Wd.items():
        v = torch.as_tensor(v)
        if ".codebook." in k:
            k = k.replace(".codebook.", "._codebook.")
        elif v.ndim == 3:
            tconv = ("upsample" in k and ".0.conv.weight" in k) or ("decoder.decoder" in k and "block.1.conv.weight" in k)
            v = v.permute(2, 0, 1) if tconv else v.permute(0, 2, 1)           # MLX [out,k,in] -> torch [in,out,k] / [out,in,k]
        k = k.replace(".layers.0.", ".0.").replace(".layers.1.", ".1.") if k.startswith("decoder.upsample.") else k
        pt[k] = v.contiguous()
    save_file(pt, str(d / "speech_tokenizer" / "model.safetensors"))
    dj = {f: (list(getattr(dcfg, f)) if isinstance(getattr(dcfg, f), tuple) else getattr(dcfg, f)) for f in mas.Qwen3TTSDecoderConfiguration.__dataclass_fields__}
    (d / "speech_tokenizer" / "config.json").write_text(json.dumps({"decoder_config": dj}))
    dev = mas.Qwen3TTSModel.from_pretrained(str(d))
    lib = mas._lib.lib()
    assert [lib.mis_tts_native_quant_bits(lib.mis_qwen3tts_talker(dev._h), r) for r in range(5)] == [8] * 5
    rng = np.random.default_rng(8)
    prompts = [_prompt(ocfg, rng, 6, 3), _prompt(ocfg, rng, 5, 1)]
    gp = mas.Qwen3TTSGenerateParameters(max_tokens=5, temperature=0.9, top_k=20, seed=3)
    a, b = ref.generate_batch(prompts, gp, return_codes=True), dev.generate_batch(prompts, gp, return_codes=True)
    for r in range(2):
        assert np.array_equal(a[1][r], b[1][r]) and np.array_equal(a[0][r], b[0][r])
    with pytest.raises(mas.AudioGenerationError):
        mas.Qwen3TTSModel.from_pretrained("mlx-community/Qwen3-TTS-12Hz-0.6B-Base-8bit")          # no network: local directories only


def test_generate_beyond_300_frames_decodes_like_decode_chunk_and_trims_to_valid_len():
    """Non-streaming generate = decodeChunk (Qwen3TTS.swift:214-231): streamingDecode(chunkTokens: 300) - carried-state steps of 300
    frames from a fresh state, i.e. the reference's doubled block bias right behind frame 300 / 600 - followed by the validLen trim
    (frames whose first code is > 0, times the upsample rate).  Two ragged rows past 300 frames (one past 600 would be slow on the
    oracle: 330 and 305); the oracle decodes the ENGINE's codes the same way.  In exact mode the same call equals the
    whole-sequence decode."""
    # EOS out of reach: greedy decoding, and the EOS row of the codec head is zeroed (logit 0 against a maximum of a few sigma)
    ocfg = oq.TINY
    W = oq.make_synthetic_weights(ocfg)
    Wd = oq.make_synthetic_decoder_weights(ocfg.decoder)
    W = {k: torch.as_tensor(v).clone() for k, v in W.items()}
    W["codec_head.weight"][ocfg.codec_eos_token_id] = 0
    allw = {("talker." + k): v for k, v in W.items()}
    allw.update(Wd)
    cfg, dev, odec = ocfg, mas.Qwen3TTSModel.from_weights(_host_cfg(ocfg), allw), oq.SpeechDecoderOracle(ocfg.decoder, Wd)
    rng = np.random.default_rng(8)
    prompts = [_prompt(cfg, rng, 6, 2), _prompt(cfg, rng, 4, 3)]
    gp = mas.Qwen3TTSGenerateParameters(max_tokens=330, temperature=0.0, repetition_penalty=1.05, seed=21)
    rows = [mas.PreparedPrompt(p.text_ids, p.codec_ids, p.trailing_ids, t) for p, t in zip(prompts, (55, 51))]   # caps 6 x 55 = 330, 306
    pcm, codes = dev.generate_batch(rows, gp, return_codes=True)
    up = cfg.decoder.total_upsample
    assert max(len(c) for c in codes) > 300, [len(c) for c in codes]        # (rows may end early on EOS; one at least runs past 300)
    for r in range(2):
        n = len(codes[r])
        ref = _oracle_stream(odec, codes[r].T[None], [300])[0]
        valid = int((codes[r][:, 0] > 0).sum()) * up
        want = valid if 0 < valid < n * up else n * up
        assert len(pcm[r]) == want, (r, len(pcm[r]), want)
        assert np.abs(pcm[r] - ref[:want]).max() <= 5e-4 * max(np.abs(ref).max(), 1e-3)
        if n > 300:
            whole = odec.decode(codes[r].T[None])[0]
            assert np.abs(ref[300 * up:300 * up + 64] - whole[300 * up:300 * up + 64]).max() > 0     # the boundary quirk is really there
    dev.set_stream_exact(True)
    try:
        pcm_x, codes_x = dev.generate_batch(rows, gp, return_codes=True)
    finally:
        dev.set_stream_exact(False)
    for r in range(2):
        assert np.array_equal(codes_x[r], codes[r])
        whole = odec.decode(codes[r].T[None])[0]
        assert np.abs(pcm_x[r] - whole[:len(pcm_x[r])]).max() <= 5e-4 * max(np.abs(whole).max(), 1e-3)


def test_group_of_two_logical_shards_streams_per_replica_and_returns_the_unsharded_rows():
    """mis_qwen3tts_group_generate (BASELINE configs[4]: streaming generateStream sharded over the GPUs of a node): three rows over
    two replicas (same weights, one GPU).  Codes and samples equal the single-handle call; with a callback every replica streams its
    own rows' chunks under their GLOBAL row index, and a row's chunks concatenate to its pcm."""
    cfg, dev, _, _ = _pair()
    _, dev2, _, _ = _pair()
    rng = np.random.default_rng(4)
    prompts = [_prompt(cfg, rng, 7, 2), _prompt(cfg, rng, 5, 4), _prompt(cfg, rng, 6, 1)]
    gp = mas.Qwen3TTSGenerateParameters(max_tokens=9, temperature=0.9, top_k=50, repetition_penalty=1.05, seed=5)
    pcm, codes = dev.generate_batch(prompts, gp, return_codes=True)
    pcm2, codes2 = dev.generate_batch(prompts, gp, return_codes=True, replicas=[dev, dev2])
    for r in range(3):
        assert np.array_equal(codes[r], codes2[r]) and np.array_equal(pcm[r], pcm2[r]), r
    got = {0: [], 1: [], 2: []}
    pcm3 = dev.generate_batch(prompts, gp, streaming_interval=0.32, on_audio=lambda row, a: got[row].append(a), replicas=[dev, dev2])
    single = {0: [], 1: [], 2: []}
    pcm4 = dev.generate_batch(prompts, gp, streaming_interval=0.32, on_audio=lambda row, a: single[row].append(a))
    for r in range(3):
        assert len(got[r]) == len(single[r]) >= 2
        assert np.array_equal(np.concatenate(got[r]), pcm3[r]) and np.array_equal(pcm3[r], pcm4[r]), r


def test_batched_prompt_prefill_matches_the_position_by_position_graph(monkeypatch):
    """The talker's prompt as one [positions x rows] pass over input embeddings (tts_internal_prefill_rows -> lm_prefill.hip, causal
    attention as two launches over (position, row) pairs) against the position-by-position graph (MIS_PREFILL_SEQ=1): greedy codes of
    ragged rows (instruct prefixes of different lengths) agree frame for frame, and both agree with the oracle under teacher forcing
    wherever its margin is not a rounding matter (the existing frame-loop tests run on the batched path by default)."""
    cfg, dev, orc, _ = _pair()
    rng = np.random.default_rng(41)
    prompts = [_prompt(cfg, rng, 5 + 9 * i, 3) for i in range(4)]               # 8 .. 35 prompt positions
    gp = mas.Qwen3TTSGenerateParameters(max_tokens=5, temperature=0.0)
    monkeypatch.setenv("MIS_PREFILL_SEQ", "0")
    a = dev.generate_codes(prompts, gp)
    monkeypatch.setenv("MIS_PREFILL_SEQ", "1")
    b = dev.generate_codes(prompts, gp)
    params = dict(temperature=0.0, top_p=1.0, top_k=0, repetition_penalty=1.05, min_p=0.0, seed=0)
    V = cfg.talker.vocab_size
    for r, p in enumerate(prompts):
        _, tl = orc.generate_row(p.text_ids, p.codec_ids, p.trailing_ids, params, row=r, max_frames=len(a[r]), forced_codes=a[r])
        lg = tl[0].copy()
        lg[V - 1024:V] = -np.inf
        lg[cfg.codec_eos_token_id] = tl[0][cfg.codec_eos_token_id]
        top = np.sort(lg)[-2:]
        if top[1] - top[0] > 0.05 * max(1.0, abs(top[1])):                      # frame 0 is the prefill's own logits
            assert int(np.argmax(lg)) == int(a[r][0][0]) == int(b[r][0][0]), r
    same = np.mean([np.array_equal(x[:min(len(x), len(y))], y[:min(len(x), len(y))]) for x, y in zip(a, b)])
    assert same >= 0.75, same            # a near-tie may flip a code between the two float32 summation orders; most rows are identical
