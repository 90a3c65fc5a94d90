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


def _pair(ocfg=oq.TINY, eos=None):
    if eos is not None:
        ocfg = oq.Qwen3TTSConfig(**{**ocfg.__dict__, "codec_eos_token_id": eos})
    W = oq.make_synthetic_weights(ocfg)
    Wd = oq.make_synthetic_decoder_weights(ocfg.decoder)
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


@pytest.mark.parametrize("ocfg", [oq.TINY, oq.TINY_PROJ], ids=["same-width", "mtp-projection"])
def test_frame_loop_greedy_under_teacher_forcing(ocfg):
    cfg, dev, olm, _ = _pair(ocfg)
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
    # end to end: pcm == speech-tokenizer oracle applied to the engine's codes; chunks concatenate to the pcm
    chunks = {0: [], 1: []}
    pcm, codes = dev.generate_batch(prompts, gp, return_codes=True, streaming_interval=0.2, on_audio=lambda r, x: chunks[r].append(x))
    for r in range(2):
        assert np.array_equal(codes[r], a[r])
        ref = odec.decode(codes[r].T[None])[0]
        assert pcm[r].shape == ref.shape == (8 * cfg.decoder.total_upsample,)
        assert np.abs(pcm[r] - ref).max() <= 5e-4 * max(np.abs(ref).max(), 1e-3)
        assert [len(x) for x in chunks[r]] == [2 * cfg.decoder.total_upsample] * 4      # 0.2 s * 12.5 Hz = 2 frames per chunk
        assert np.array_equal(np.concatenate(chunks[r]), pcm[r])
