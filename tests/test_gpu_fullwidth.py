"""-m gpu: parity against the oracle AT THE WIDTHS THAT ARE BENCHMARKED (VERDICT r01 "what's weak" 2).

The small-config tests elsewhere exercise every code path cheaply; the engines pick different instantiations at the
benchmark widths (Orpheus-3B: MT=2, R=2, KSB=4, split-K S = 3 / 2 / 8 from gemm_choose_split, 160 / 96 n-tile groups, K = 8192
tails, 24 / 8 heads, 156 940-wide lm_head; Whisper-large-v3: 1280 x 20 heads, 128 mels, 5120 ffn; Qwen3-TTS-0.6B: 1024-wide
talker / predictor, the real speech-tokenizer decoder; SNAC at 96 groups; DAC / EnCodec 24 kHz dims).  Layer COUNT is cut (1-2
layers) so the oracle finishes in seconds; every per-layer shape is the full model's.
Tolerances are those of the small-config tests (stated per test); the observed errors are recorded with gpu_util.record()."""
import numpy as np
import pytest
import torch

import mlx_audio_swift_amd as mas
from gpu_util import codec_exact_f32, lm_host_config, logits_errors, record, rms
from oracle import dac as od
from oracle import encodec as oe
from oracle import llama as ollama
from oracle import qwen3tts as oq
from oracle import snac as osnac
from oracle import whisper as ow

pytestmark = pytest.mark.gpu


def test_orpheus_3b_width_teacher_forced_b32_contexts_40_400_705():
    """Orpheus-3B per-layer shapes (d 3072, ffn 8192, 24/8 heads x 128, V 156 940, tied head), 2 layers, batch 32 = the bench's
    GEMM instantiations; ragged contexts so that attention runs 2 / 13 / 23 key tiles (incl. the second tile pair round and the
    patched new-key tile beyond 512).  Tolerance: logits max <= 0.016 max|ref|, rms <= 0.008 rms(ref) (tests/test_gpu_lm.py; observed 0.0075 / 0.0069)."""
    cfg = ollama.LlamaConfig(num_hidden_layers=2)                       # every other field = ORPHEUS_3B
    W = ollama.make_synthetic_weights(cfg, seed=4321)
    oracle = ollama.LlamaOracle(cfg, W, round="bf16")
    dev = mas.LlamaTTSModel.synthetic(lm_host_config(cfg), seed=4321)   # same generator on the device (equality: test_gpu_lm.py)
    assert mas._lib.lib().mis_debug_choose_split(160, 96, 4, 8) == 3    # the bench's qkv split (5120/16/2 items, 3072/32 k-tiles)
    B = 32
    lens = [705, 705, 400, 400] + [36 + (b % 9) for b in range(4, B)]
    rng = np.random.default_rng(21)
    rows = [np.concatenate([[128259], rng.integers(0, 128000, n - 1)]).astype(np.int32) for n in lens]
    checks = [sorted({n - 1, *[p for p in (39, 399) if p < n - 1]}) for n in lens]          # positions whose logits are compared
    want_steps = sorted({p for c in checks for p in c})
    dev.lm_reset(B, 768)
    got = {}
    for t in range(max(lens)):
        ids = np.asarray([r[t] if t < len(r) else 0 for r in rows], np.int32)
        act = np.asarray([1 if t < len(r) else 0 for r in rows], np.uint8)
        if t in want_steps:
            lg = dev.lm_forward(ids, act)
            for b in range(B):
                if t in checks[b]:
                    got[(b, t)] = lg[b].copy()
        else:
            dev.lm_forward(ids, act, want_logits=False)
    oracle.reset(B)
    ref = oracle.forward(rows, logit_positions=checks)
    worst = (0.0, 0.0)
    for b in range(B):
        d = np.stack([got[(b, p)] for p in checks[b]])
        r = ref[b].numpy()
        assert d.shape == r.shape
        e_max, e_rms, n_sure, agree = logits_errors(d, r)
        worst = (max(worst[0], e_max), max(worst[1], e_rms))
        assert e_max <= 0.016 and e_rms <= 0.008, (b, e_max, e_rms)
        assert agree, b
        assert torch.equal(torch.from_numpy(d), torch.from_numpy(d).bfloat16().float())          # bf16-valued logits
    record("orpheus3b_width_b32_ctx705", logits_max_rel=worst[0], logits_rms_rel=worst[1], tol_max=0.016, tol_rms=0.008)
    # the same rows alone (B = 1 -> MT = 1 instantiation) give bit-identical logits: batching is exact
    dev.lm_reset(1, 768)
    for t in range(lens[4]):
        one = dev.lm_forward(rows[4][t:t + 1], want_logits=(t == lens[4] - 1))
    assert np.array_equal(one[0], got[(4, lens[4] - 1)])


def test_whisper_large_v3_width_encoder_and_decoder_layer():
    """large-v3 per-layer shapes (d 1280, 20 heads x 64, ffn 5120, 128 mels, V 51 866), 1 + 1 layers.
    Tolerance (about twice the errors observed on MI355X: 0.0090 / 0.0034 / 0.0048 / 0.0041): encoder max <= 0.02, rms <= 0.008; decoder
    logits max <= 0.012, rms <= 0.009."""
    cfg = ow.WhisperConfig(vocab_size=51866, num_mel_bins=128, d_model=1280, encoder_layers=1, encoder_attention_heads=20,
                           encoder_ffn_dim=5120, decoder_layers=1, decoder_attention_heads=20, decoder_ffn_dim=5120)
    W = ow.make_synthetic_weights(cfg, seed=777)
    oracle = ow.WhisperOracle(cfg, W, round="bf16")
    dev = mas.WhisperModel.from_weights(mas.WhisperConfig(**{k: getattr(cfg, k) for k in mas.WhisperConfig.__dataclass_fields__}), W)
    B = 2
    feats = (np.random.default_rng(1).standard_normal((B, 3000, 128)) * 0.5).astype(np.float32)
    oracle.reset(B)
    enc_ref = oracle.encode(feats)
    enc = dev.encode(feats)
    e = [(float(np.abs(enc[b] - enc_ref[b].numpy()).max() / np.abs(enc_ref[b].numpy()).max()),
          rms(enc[b], enc_ref[b].numpy()) / float(np.sqrt(np.mean(enc_ref[b].numpy().astype(np.float64) ** 2)))) for b in range(B)]
    assert max(x[0] for x in e) <= 0.02 and max(x[1] for x in e) <= 0.008, e
    toks = np.random.default_rng(2).integers(0, cfg.vocab_size, (B, 40))
    dev.decoder_reset()
    got = [dev.decoder_forward(toks[:, t]) for t in range(40)]
    ref = oracle.decode([toks[0], toks[1]])
    worst = (0.0, 0.0)
    for b in range(B):
        d = np.stack([g[b] for g in got]); r = ref[b].numpy()
        e_max, e_rms, n_sure, agree = logits_errors(d, r)
        worst = (max(worst[0], e_max), max(worst[1], e_rms))
        assert e_max <= 0.012 and e_rms <= 0.009 and agree, (b, e_max, e_rms)
    record("whisper_large_v3_width", enc_max_rel=max(x[0] for x in e), enc_rms_rel=max(x[1] for x in e), dec_logits_max_rel=worst[0],
           dec_logits_rms_rel=worst[1], tol=[0.02, 0.008, 0.012, 0.009])


def test_qwen3tts_06b_width_frame_loop_and_real_decoder():
    """Qwen3-TTS-0.6B widths: talker 2 layers / predictor 1 layer at hidden 1024, 16/8 heads x 128, ffn 3072, codec vocab 3072,
    16 code groups; the speech-tokenizer decoder at its real dimensions (8 transformer layers, 1536-wide vocoder).
    Tolerance (about twice the observed 0.0051 / 9.8e-5): greedy choice within 0.012 max|logit| of the oracle's maximum; waveform 2e-4."""
    from test_gpu_qwen3tts import _host_cfg, _prompt
    base = oq.Qwen3TTSConfig()
    ocfg = oq.Qwen3TTSConfig(**{**base.__dict__,
                                "talker": ollama.LlamaConfig(**{**base.talker.__dict__, "num_hidden_layers": 2}),
                                "predictor": ollama.LlamaConfig(**{**base.predictor.__dict__, "num_hidden_layers": 1}),
                                "text_vocab_size": 4096, "tts_pad_token_id": 4000, "tts_bos_token_id": 4001, "tts_eos_token_id": 4002})
    W = oq.make_synthetic_weights(ocfg)
    Wd = oq.make_synthetic_decoder_weights(ocfg.decoder)
    allw = {("talker." + k): v for k, v in W.items()}
    allw.update(Wd)
    dev = mas.Qwen3TTSModel.from_weights(_host_cfg(ocfg), allw)
    olm, odec = oq.Qwen3TTSOracle(ocfg, W), oq.SpeechDecoderOracle(ocfg.decoder, Wd)
    cfg = ocfg
    rng = np.random.default_rng(2)
    prompts = [_prompt(cfg, rng, 9, 3), _prompt(cfg, rng, 5, 1)]
    gp = mas.Qwen3TTSGenerateParameters(max_tokens=4, temperature=0.0, repetition_penalty=1.05, seed=1)
    codes = dev.generate_codes(prompts, gp)
    assert [c.shape for c in codes] == [(4, 16)] * 2
    suppress = [t for t in range(cfg.talker.vocab_size - 1024, cfg.talker.vocab_size) if t != cfg.codec_eos_token_id]
    pr = dict(temperature=0.0, top_p=1.0, top_k=0, repetition_penalty=1.05, min_p=0.0, seed=1)
    worst = 0.0
    for b, p in enumerate(prompts):
        olm.talker.reset(1)
        x = olm.position_embeds(p.text_ids, p.codec_ids)
        trailing = olm.text_embed(p.trailing_ids)
        pad = olm.text_embed([cfg.tts_pad_token_id])[0]
        gen0 = []
        for f in range(4):
            lg = olm.talker.forward_embeds(0, x)[-1].numpy()
            hidden = olm.talker.last_hidden[-1]
            l = lg.copy(); l[suppress] = -np.inf
            pen = oq.synth.bf16_round(np.asarray([1.05], np.float32))[0]
            for t in set(gen0):
                l[t] = l[t] * pen if l[t] < 0 else l[t] / pen
            c0 = int(codes[b][f, 0])
            gap = float(l.max() - l[c0]) / float(np.abs(lg).max())
            worst = max(worst, gap)
            assert c0 < cfg.talker.vocab_size - 1024 and gap <= 0.012, (b, f, gap)
            _, plog = olm.predictor_codes(hidden, c0, pr, b, f, forced=codes[b][f], want_logits=True)
            for i, pl in enumerate(plog):
                ci = int(codes[b][f, i + 1])
                g2 = float(pl.max() - pl[ci]) / float(np.abs(pl).max())
                worst = max(worst, g2)
                assert g2 <= 0.012, (b, f, i, g2)
            te = trailing[f] if f < trailing.shape[0] else pad
            x = olm.next_input(te, [int(v) for v in codes[b][f]])[None]
            gen0.append(c0)
    d = cfg.decoder
    cd = rng.integers(0, d.codebook_size, (2, d.num_quantizers, 13)).astype(np.int32)
    ref = odec.decode(cd)
    got = dev.decode_codes(cd)
    werr = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-3))
    assert got.shape == ref.shape == (2, 13 * 1920) and werr <= 2e-4, werr
    with codec_exact_f32():                                # the same decode on the exact-f32 kernels: the split-bf16 path ran, and is close
        ex = dev.decode_codes(cd)
    serr = float(np.abs(got - ex).max() / max(np.abs(ex).max(), 1e-3))
    assert not np.array_equal(got, ex) and serr <= 2.5e-4, serr       # observed 1.0e-4 of the clipped full-scale waveform
    record("qwen3tts_06b_width", greedy_gap_rel=worst, tol_gap=0.012, decoder_wave_max_rel=werr, tol_wave=2e-4, split_bf16_vs_exact_f32_max_rel=serr)


def test_snac_24khz_one_row_96_groups():
    """SNAC 24 kHz at the bench's decode length (96 groups = 8.192 s per row), explicit noise.  Tolerance: RMS <= 1e-4 (north_star)."""
    from gpu_util import snac_pair
    ocfg, oracle, dev = snac_pair({})
    codes = osnac.synthetic_codes(ocfg, 1, 96, seed=5)
    noise = osnac.synthetic_noise(ocfg, 1, 96, seed=6)
    ref = oracle.decode(codes, noise)
    got = dev.decode(codes, noise)
    e = rms(got, ref)
    assert got.shape == ref.shape == (1, 1, 96 * 2048) and e < 1e-4, e          # the north-star bound
    assert e < 2e-5, e                                                          # twice the 8.9e-6 observed on MI355X
    # the same row inside a batch of 32 (the bench batch): bit-identical to the single-row decode
    codes32 = [np.repeat(c, 32, axis=0) for c in codes]
    noise32 = [np.repeat(n, 32, axis=0) for n in noise]
    got32 = dev.decode(codes32, noise32)
    assert np.array_equal(got32[0], got[0]) and np.array_equal(got32[31], got[0])
    with codec_exact_f32():
        ex = dev.decode(codes, noise)
    es = rms(got, ex)
    assert not np.array_equal(got, ex) and es < 3e-5 and rms(ex, ref) < 1e-4, es
    record("snac_24khz_96_groups", wave_rms=e, wave_max=float(np.abs(got - ref).max()), tol_rms=1e-4, split_bf16_vs_exact_f32_rms=es,
           exact_f32_wave_rms=rms(ex, ref))


def test_dac_24khz_and_encodec_24khz_real_dims():
    """Descript DAC 24 kHz (decoder_dim 1536, rates 8/5/4/2, 32 codebooks x 1024 x 8) and EnCodec 24 kHz (32 filters, hidden 128,
    2-layer LSTM, 32 quantizers): waveform max error <= 1e-4 max|ref| (observed 5.0e-5 / 1.2e-5)."""
    c = od.DacConfig(n_codebooks=32, sample_rate=24000)
    W = od.make_synthetic_weights(c)
    stored = {k.replace(".outProj.", ".out_proj.").replace("decoder.model.", "decoder.model.layers."): v for k, v in W.items()}
    hcfg = mas.DescriptDACConfig(**{k: getattr(c, k) for k in mas.DescriptDACConfig.__dataclass_fields__})
    orc, dev = od.DacOracle(c, W), mas.DescriptDAC.from_weights(hcfg, stored)
    codes = np.random.default_rng(0).integers(0, 1024, (2, 32, 25)).astype(np.int32)
    ref = orc.decode_from_codes(codes)
    got = dev.decode_from_codes(codes)
    e_dac = float(np.abs(got - ref).max() / np.abs(ref).max())
    assert got.shape == ref.shape == (2, od.num_samples(c, 25)) and e_dac <= 1e-4, e_dac
    with codec_exact_f32():
        ex = dev.decode_from_codes(codes)
    s_dac = float(np.abs(got - ex).max() / np.abs(ex).max())
    assert not np.array_equal(got, ex) and s_dac <= 1e-4, s_dac
    ec = oe.EncodecConfig()
    We = oe.make_synthetic_weights(ec)
    fields = {k: getattr(ec, k) for k in mas.EncodecConfig.__dataclass_fields__ if hasattr(ec, k)}
    eo, edev = oe.EncodecOracle(ec, We), mas.Encodec.from_weights(mas.EncodecConfig(**fields), We)
    codes = np.random.default_rng(1).integers(0, 1024, (2, ec.num_quantizers, 75)).astype(np.int32)       # 1 s of audio
    ref = eo.decode_frame(codes)
    got = edev.decode_frame(codes)
    e_enc = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-3))
    assert got.shape == ref.shape == (2, 75 * 320) and e_enc <= 1e-4, e_enc
    with codec_exact_f32():
        ex = edev.decode_frame(codes)
    s_enc = float(np.abs(got - ex).max() / max(np.abs(ex).max(), 1e-3))
    assert s_enc <= 1e-4, s_enc
    record("dac_encodec_24khz_real_dims", dac_wave_max_rel=e_dac, encodec_wave_max_rel=e_enc, tol=1e-4, dac_split_bf16_vs_exact_f32_max_rel=s_dac,
           encodec_split_bf16_vs_exact_f32_max_rel=s_enc)
