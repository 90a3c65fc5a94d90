"""-m gpu: the other BASELINE configurations at their REAL depth and dimensions (VERDICT r03 "what's missing" 1).

test_gpu_depth.py does this for the benchmarked Orpheus-3B; here:
(a) Whisper-large-v3 (configs[3]): 32 encoder + 32 decoder layers, d 1280, 20 heads, 128 mels, V 51 866 - one 30 s window (3000 frames ->
    1500 encoder positions), 24 teacher-forced decoder positions, against oracle/whisper.py; the same comparison at 2 + 2 and 8 + 8
    layers puts the growth with depth on file, and the oracle's own float64-accumulation floor is measured beside it (as in
    test_gpu_depth.py: two exact realisations of one specification differ because summation order flips bf16 roundings).
(b) Qwen3-TTS-0.6B (configs[4]): talker 28 layers / code predictor 5 layers at hidden 1024, bf16 AND as the 8-bit checkpoint the
    config names (MLX affine quantisation, group 64, streamed natively): 4 frames, the engine's greedy choice of every code group
    against the oracle's logits under teacher forcing.
(c) Soprano-80M (configs[1]): the Vocos decoder at its real dimensions - 8 ConvNeXt layers, dim 768 / 2304, n_fft 2048, hop 512,
    hidden 512 (SopranoConfig.swift:158-167) - waveform against oracle/soprano.py.
(d) Soprano-80M (configs[1]): the batch-1 TOKEN ENGINE (csrc/token_engine.hip, the program that runs the LM loop of configs[1]) at the
    LM's real shape - 17 layers, vocabulary 8 192 (the limit of token_engine_supports: at one XCD the output projection takes its
    second pass) - 24 prompt + 64 generated positions on 1 and 4 XCDs: logits and hidden rows against oracle/llama.py under teacher
    forcing, with the 2- and 8-layer comparison and the oracle's float64 floor beside it; and the in-launch sampler bit-exact on the
    engine's own logits at that shape.
Tolerances are stated per test and are about twice what MI355X delivered when the test was written (recorded with gpu_util.record)."""
import dataclasses
import gc

import numpy as np
import pytest
import torch

import mlx_audio_swift_amd as mas
from gpu_util import lm_host_config, logits_errors, record, rms
from oracle import llama as ollama
from oracle import qwen3tts as oq
from oracle import soprano as osop
from oracle import whisper as ow

pytestmark = pytest.mark.gpu

FLOOR_FACTOR = 2.0


class _F64Whisper(ow.WhisperOracle):
    """Same graph, same rounding points; every Linear accumulated in float64 (the noise-floor reference)."""

    def linear(self, x, p, bias=True):
        y = x.to(torch.float64) @ self.w[p + ".weight"].to(torch.float64).t()
        if bias and (p + ".bias") in self.w:
            y = y + self.w[p + ".bias"].to(torch.float64)
        return self.r(y.to(torch.float32))


def _rel(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max()), rms(a, b) / float(np.sqrt(np.mean(b ** 2)))


@pytest.mark.parametrize("depths", [pytest.param((2, 8), id="2_and_8_layers"),
                                    pytest.param((2, 8, 32), id="full_depth_32_plus_32_layers", marks=pytest.mark.slow)])
def test_whisper_large_v3_full_depth_32_plus_32_layers_and_error_growth(depths):
    """(the 32 + 32 variant needs ~140 s, most of it the CPU oracle: marked slow - run on the final tree with --runslow; the default set
    holds the device to the same floor gate at 2 + 2 and 8 + 8 layers)"""
    T = 24
    feats = (np.random.default_rng(1).standard_normal((1, 3000, 128)) * 0.5).astype(np.float32)
    toks = np.random.default_rng(2).integers(0, 50000, (1, T))
    growth = {}
    for L in depths:
        cfg = dataclasses.replace(ow.LARGE_V3, encoder_layers=L, decoder_layers=L)
        W = ow.make_synthetic_weights(cfg, seed=777)           # (the key counter runs through the layers: one dict per depth)
        o32 = ow.WhisperOracle(cfg, W, round="bf16")
        del W
        gc.collect()
        o64 = _F64Whisper.__new__(_F64Whisper)
        o64.__dict__.update(o32.__dict__)                      # shares the float32 weight dict
        dev = mas.WhisperModel.synthetic(mas.WhisperConfig(**{k: getattr(cfg, k) for k in mas.WhisperConfig.__dataclass_fields__}), seed=777)
        enc = dev.encode(feats)
        dev.decoder_reset()
        got = np.stack([dev.decoder_forward(toks[:, t])[0] for t in range(T)])           # [T, V]
        dev.close()
        del dev
        gc.collect()
        res = {}
        for name, o in (("ref", o32), ("floor", o64)):
            o.reset(1)
            e = o.encode(feats)[0].numpy()
            lg = o.decode([toks[0]])[0].numpy()
            res[name] = (e, lg)
        e_max, e_rms = _rel(enc[0], res["ref"][0])
        fe_max, fe_rms = _rel(res["floor"][0], res["ref"][0])
        d_max, d_rms, n_sure, agree = logits_errors(got, res["ref"][1])
        fd_max, fd_rms, _, _ = logits_errors(res["floor"][1], res["ref"][1])
        assert agree and n_sure > 0, L                         # greedy token wherever the oracle's margin exceeds 2x the error
        growth[L] = dict(enc_max=e_max, enc_rms=e_rms, enc_floor_max=fe_max, enc_floor_rms=fe_rms, dec_max=d_max, dec_rms=d_rms,
                         dec_floor_max=fd_max, dec_floor_rms=fd_rms)
        record(f"whisper_large_v3_depth_{L}_plus_{L}", layers=L, **growth[L], gate=f"rms: dev <= {FLOOR_FACTOR} x floor (+2e-3); absolute at 32+32")
        del o32, o64, res
        gc.collect()
    for L, g in growth.items():
        assert g["enc_rms"] <= FLOOR_FACTOR * g["enc_floor_rms"] + 2e-3, (L, g)
        assert g["dec_rms"] <= FLOOR_FACTOR * g["dec_floor_rms"] + 2e-3, (L, g)
    # absolute bounds, about twice what MI355X delivered (profiles/r04_parity_observed.json): 32 + 32 layers encoder rms 0.0113 / max 0.0299,
    # decoder logits rms 0.0139 / max 0.0164 - against the oracle's own float64 floor of 0.0112 / 0.0299 and 0.0138 / 0.0164: the device
    # sits AT the floor at every depth (8 + 8: 0.0076 vs 0.0073, 0.0086 vs 0.0086; 2 + 2: 0.0048 vs 0.0039, 0.0056 vs 0.0055)
    if 32 in growth:
        assert growth[32]["enc_rms"] <= 0.023 and growth[32]["dec_rms"] <= 0.028, growth[32]
        assert growth[32]["enc_max"] <= 0.06 and growth[32]["dec_max"] <= 0.033, growth[32]
    assert growth[8]["enc_rms"] <= 0.016 and growth[8]["dec_rms"] <= 0.018, growth[8]
    assert growth[2]["enc_rms"] <= 0.010 and growth[2]["dec_rms"] <= 0.012, growth[2]


@pytest.mark.parametrize("quant", [None, 8], ids=["bf16", "8bit-checkpoint"])
def test_qwen3tts_06b_full_depth_talker_28_predictor_5(quant):
    """Greedy choice within `tol` of the oracle's maximum for code group 0 (talker, 28 layers) and groups 1..15 (predictor, 5 layers),
    4 frames, 2 rows; bf16 weights and the 8-bit checkpoint form.  tol: 0.012 max|logit| (observed on MI355X at full depth: talker 0.0
    - the engine's choice IS the oracle's argmax in all 8 frames - predictor 0.0043 bf16 / 0.0047 8 bit; the width test at 2 + 1 layers
    observed 0.0051 with the same bound)."""
    from test_gpu_qwen3tts import _host_cfg, _prompt
    base = oq.Qwen3TTSConfig()
    assert base.talker.num_hidden_layers == 28 and base.predictor.num_hidden_layers == 5 and base.talker.hidden_size == 1024
    ocfg = oq.Qwen3TTSConfig(**{**base.__dict__, "decoder": oq.TINY.decoder, "text_vocab_size": 4096, "tts_pad_token_id": 4000,
                                "tts_bos_token_id": 4001, "tts_eos_token_id": 4002})
    W = oq.make_synthetic_weights(ocfg)
    Wd = oq.make_synthetic_decoder_weights(ocfg.decoder)
    if quant:
        from oracle import mlxquant as mq
        dev = mas.Qwen3TTSModel(_host_cfg(ocfg))
        Wo = {}
        for k, v in W.items():
            v = torch.as_tensor(v)
            if v.ndim == 2 and v.shape[1] % 64 == 0:
                wq, s, bia = mq.quantize(v.float().numpy(), 64, quant)
                s16, b16 = torch.from_numpy(s).bfloat16(), torch.from_numpy(bia).bfloat16()
                dev.set_quantized_tensor("talker." + k, wq, s16, b16, 64, quant)
                d32 = torch.from_numpy(mq.dequantize(wq, s16.float().numpy(), b16.float().numpy(), 64, quant))
                lm_linear = ("model.layers." in k and k.endswith("_proj.weight")) or k == "codec_head.weight"
                Wo[k] = d32 if lm_linear else d32.bfloat16()
            else:
                dev.set_tensor("talker." + k, v); Wo[k] = v
        for k, v in Wd.items():
            dev.set_tensor(k, v)
        dev.finalize()
        lib = mas._lib.lib()
        assert [lib.mis_tts_native_quant_bits(lib.mis_qwen3tts_talker(dev._h), r) for r in range(5)] == [quant] * 5
        olm = oq.Qwen3TTSOracle(ocfg, Wo)
    else:
        allw = {("talker." + k): v for k, v in W.items()}
        allw.update(Wd)
        dev = mas.Qwen3TTSModel.from_weights(_host_cfg(ocfg), allw)
        olm = oq.Qwen3TTSOracle(ocfg, W)
    cfg = ocfg
    rng = np.random.default_rng(2)
    prompts = [_prompt(cfg, rng, 9, 3), _prompt(cfg, rng, 5, 1)]
    F = 4
    gp = mas.Qwen3TTSGenerateParameters(max_tokens=F, temperature=0.0, repetition_penalty=1.05, seed=1)
    codes = dev.generate_codes(prompts, gp)
    assert [c.shape for c in codes] == [(F, 16)] * 2
    suppress = [t for t in range(cfg.talker.vocab_size - 1024, cfg.talker.vocab_size) if t != cfg.codec_eos_token_id]
    pr = dict(temperature=0.0, top_p=1.0, top_k=0, repetition_penalty=1.05, min_p=0.0, seed=1)
    worst_t = worst_p = 0.0
    tol = 0.012
    for b, p in enumerate(prompts):
        olm.talker.reset(1)
        x = olm.position_embeds(p.text_ids, p.codec_ids)
        trailing = olm.text_embed(p.trailing_ids)
        pad = olm.text_embed([cfg.tts_pad_token_id])[0]
        gen0 = []
        for f in range(F):
            lg = olm.talker.forward_embeds(0, x)[-1].numpy()
            hidden = olm.talker.last_hidden[-1]
            l = lg.copy(); l[suppress] = -np.inf
            pen = oq.synth.bf16_round(np.asarray([1.05], np.float32))[0]
            for t in set(gen0):
                l[t] = l[t] * pen if l[t] < 0 else l[t] / pen
            c0 = int(codes[b][f, 0])
            gap = float(l.max() - l[c0]) / float(np.abs(lg).max())
            worst_t = max(worst_t, gap)
            assert c0 < cfg.talker.vocab_size - 1024 and gap <= tol, (b, f, gap)
            _, plog = olm.predictor_codes(hidden, c0, pr, b, f, forced=codes[b][f], want_logits=True)
            for i, pl in enumerate(plog):
                ci = int(codes[b][f, i + 1])
                g2 = float(pl.max() - pl[ci]) / float(np.abs(pl).max())
                worst_p = max(worst_p, g2)
                assert g2 <= tol, (b, f, i, g2)
            te = trailing[f] if f < trailing.shape[0] else pad
            x = olm.next_input(te, [int(v) for v in codes[b][f]])[None]
            gen0.append(c0)
    record(f"qwen3tts_06b_depth_28_5_{'8bit' if quant else 'bf16'}", talker_greedy_gap_rel=worst_t, predictor_greedy_gap_rel=worst_p, tol_gap=tol)


def test_soprano_80m_decoder_real_dimensions():
    """Vocos decoder at Soprano-1.1's dimensions: hidden 512 -> 768-wide, 8 ConvNeXt layers (2304 intermediate, depthwise kernel 3),
    input kernel 1, upscale 4, ISTFT head n_fft 2048 / hop 512 (2 x 1025 spectral rows per frame).  Waveform max |err| <= 4e-5 max|ref|
    (observed 1.04e-5 on MI355X; the small-config decoder tests allow 2e-4)."""
    ocfg = osop.SopranoDecoderConfig()
    assert (ocfg.decoder_num_layers, ocfg.decoder_dim, ocfg.decoder_intermediate_dim, ocfg.n_fft, ocfg.hop_length, ocfg.hidden_size) == (8, 768, 2304, 2048, 512, 512)
    LM = dataclasses.replace(ollama.TINY_QWEN3, hidden_size=512)             # the decoder's input width; the LM itself is not run here
    dec = {k: getattr(ocfg, k) for k in ("decoder_num_layers", "decoder_dim", "decoder_intermediate_dim", "hop_length", "n_fft", "upscale",
                                         "input_kernel", "dw_kernel", "token_size")}
    cfg = mas.SopranoConfiguration(hidden_size=LM.hidden_size, num_hidden_layers=LM.num_hidden_layers, intermediate_size=LM.intermediate_size,
                                   num_attention_heads=LM.num_attention_heads, num_key_value_heads=LM.num_key_value_heads, head_dim=LM.head_dim,
                                   vocab_size=LM.vocab_size, rms_norm_eps=LM.rms_norm_eps, rope_theta=LM.rope_theta, tie_word_embeddings=False,
                                   stop_token_id=3, **dec)
    Wd = osop.make_synthetic_weights(ocfg, seed=99)
    Wl = ollama.make_synthetic_weights(LM, seed=4321)
    W = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in Wl.items()}
    W.update(Wd)
    dev = mas.SopranoModel.from_weights(cfg, W)
    odec = osop.SopranoDecoderOracle(ocfg, Wd)
    rng = np.random.default_rng(0)
    worst = 0.0
    for B, L in [(1, 12), (2, 5)]:
        hid = rng.standard_normal((B, L, cfg.hidden_size)).astype(np.float32)
        ref = odec.decode(hid)
        got = dev.decode(hid)
        assert got.shape == ref.shape == (B, cfg.upscale * (L - 1) * cfg.hop_length)
        e = float(np.abs(got - ref).max() / np.abs(ref).max())
        worst = max(worst, e)
        assert e <= 4e-5, (B, L, e)
    record("soprano_80m_decoder_real_dims", wave_max_rel=worst, tol=4e-5)


class _F64Llama(ollama.LlamaOracle):
    """Same graph, same rounding points; every contraction accumulated in float64 (the noise-floor reference)."""

    def linear(self, x, w):
        return self.r((x.to(torch.float64) @ w.to(torch.float64).t()).to(torch.float32))


SOPRANO_80M_LM = ollama.LlamaConfig(hidden_size=512, num_hidden_layers=17, intermediate_size=2304, num_attention_heads=4, num_key_value_heads=1,
                                    head_dim=128, vocab_size=8192, rope_theta=10000.0, rope_scaling=None, tie_word_embeddings=False, qk_norm=True,
                                    rope_plain=True, rms_norm_eps=1e-6)


def test_token_engine_at_soprano_80m_depth_17_layers_v8192_and_error_growth():
    """the engine had parity tests at 2 layers / V = 1 200 only.  Here the shape bench.py and the product run:
    69 cross-CU edges per position through 17 layers, V = 8 192 (1 XCD: two output-projection passes).  Laboratory form (arg-max after
    every position, logits and hidden rows of all 88 positions), oracle teacher-forced with the engine's own ids.  Gate like
    test_gpu_depth.py: rms error <= FLOOR_FACTOR x the oracle's own float64-accumulation floor (+1e-3), absolute bounds at 17 layers;
    greedy ids equal wherever the oracle's margin exceeds twice the error; 1 and 4 XCDs bit-identical."""
    full = SOPRANO_80M_LM
    W = ollama.make_synthetic_weights(full, seed=4321)              # layer keys do not depend on the layer count
    o32 = ollama.LlamaOracle(full, W, round="bf16")
    del W
    o64 = _F64Llama.__new__(_F64Llama)
    o64.__dict__.update(o32.__dict__)
    rng = np.random.default_rng(23)
    prompt = rng.integers(0, full.vocab_size, 24).astype(np.int32)
    n_new = 64
    growth = {}
    for L in (2, 8, 17):
        cfg = dataclasses.replace(full, num_hidden_layers=L)
        dev = mas.LlamaTTSModel.synthetic(lm_host_config(cfg), seed=4321)
        outs = {x: dev.debug_token_engine(prompt, n_new, xcds=x, want_logits=True, want_hidden=True) for x in (1, 4)}
        del dev
        gc.collect()
        assert np.array_equal(outs[1]["logits"], outs[4]["logits"]) and np.array_equal(outs[1]["next_tokens"], outs[4]["next_tokens"]), L
        assert np.array_equal(outs[1]["hidden"], outs[4]["hidden"]), L
        out = outs[1]
        nxt = out["next_tokens"]
        assert np.array_equal(out["logits"].argmax(1), nxt)
        seq = np.concatenate([prompt, nxt[len(prompt) - 1:len(prompt) - 1 + n_new]]).astype(np.int32)
        res = {}
        for name, o in (("ref", o32), ("floor", o64)):
            o.cfg = cfg
            o.reset(1)
            lg = o.forward([seq])[0].numpy()
            res[name] = (lg, o.last_hidden.numpy())
        e_max, e_rms, n_sure, agree = logits_errors(out["logits"], res["ref"][0])
        f_max, f_rms, _, _ = logits_errors(res["floor"][0], res["ref"][0])
        h_max, h_rms = _rel(out["hidden"], res["ref"][1])
        hf_max, hf_rms = _rel(res["floor"][1], res["ref"][1])
        # the generated half alone (contexts 24 .. 87): where an error fed back through the engine's own ids would show
        g_rms = rms(out["logits"][len(prompt):], res["ref"][0][len(prompt):]) / float(np.sqrt(np.mean(res["ref"][0][len(prompt):].astype(np.float64) ** 2)))
        assert agree and n_sure > 0, (L, n_sure)
        growth[L] = dict(dev_max=e_max, dev_rms=e_rms, floor_max=f_max, floor_rms=f_rms, hidden_max=h_max, hidden_rms=h_rms,
                         hidden_floor_max=hf_max, hidden_floor_rms=hf_rms, dev_rms_generated_positions=g_rms, rows_with_a_sure_argmax=n_sure)
        record(f"token_engine_soprano80m_depth_{L}_layers_v8192", layers=L, **growth[L], ms_per_position_1xcd=outs[1]["ms"] / len(seq),
               ms_per_position_4xcd=outs[4]["ms"] / len(seq), gate=f"rms: dev <= {FLOOR_FACTOR} x floor (+1e-3); absolute at 17 layers")
    for L, g in growth.items():
        assert g["dev_rms"] <= FLOOR_FACTOR * g["floor_rms"] + 1e-3, (L, g)
        assert g["hidden_rms"] <= FLOOR_FACTOR * g["hidden_floor_rms"] + 1e-3, (L, g)
    # absolute bounds at the shipped depth (the per-row bounds of every LM test at 2 layers: max 0.016 / rms 0.008; the error grows like the
    # floor does, ~ sqrt(layers): 17 layers ~ 3 x 2 layers)
    assert growth[2]["dev_rms"] <= 0.008 and growth[2]["dev_max"] <= 0.016, growth[2]
    assert growth[17]["dev_rms"] <= 0.03 and growth[17]["dev_max"] <= 0.03, growth[17]
    assert growth[17]["hidden_rms"] <= 0.03, growth[17]


@pytest.mark.parametrize("xcds", [1, 4])
def test_token_engine_sampler_bit_exact_at_soprano_80m_shape(xcds):
    """The generate form at 17 layers / V = 8 192: every id the launch chose (mis-sampler-v1 behind the Soprano penalty: edges 5-7, 1 024
    tile-mass granules; at one XCD each worker owns ids of BOTH output-projection passes) is oracle/sampler.py's on the logits row it was
    drawn from, and temperature 0 is the arg-max behind the penalty."""
    from oracle import sampler as osamp
    cfg = SOPRANO_80M_LM
    dev = mas.LlamaTTSModel.synthetic(lm_host_config(cfg), seed=4321)
    prompt = np.random.default_rng(29).integers(0, cfg.vocab_size, 24).astype(np.int32)
    n_new = 64
    for temp in (0.0, 0.3):
        gp = mas.GenerateParameters(max_tokens=n_new, temperature=temp, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30, seed=9,
                                    row_offset=0, sampler_flavor=1)
        out = dev.debug_token_engine(prompt, n_new, xcds=xcds, want_logits=True, sampling=gp)
        assert out["chosen"] == n_new
        toks = out["next_tokens"][len(prompt) - 1:len(prompt) - 1 + n_new]
        for k in range(n_new):
            l = osop.soprano_repetition_penalty(out["logits"][k], list(toks[:k])[-30:], 1.5)
            assert toks[k] == osamp.sample(l, temp, 1.0, 9, 0, k), (temp, k)
