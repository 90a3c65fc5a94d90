"""-m gpu: Qwen3-TTS in-context voice cloning (VERDICT r02 "what's missing" 2).

(a) the nFft-1024 log-mel of extractSpeakerEmbedding vs oracle/mel.py;
(b) the speaker encoder (ECAPA-TDNN) vs oracle/ecapa.py: every block, mfa, pooled statistics and the x-vector, at a small width and at
    the checkpoint's width (512 / 1536 channels, 128 mels);
(c) the speech-tokenizer encoder (Mimi) vs oracle/mimi_encoder.py: SEANet / transformer / downsampled latent within float32 tolerance,
    codes equal wherever the oracle's nearest-code decision has a margin, at a small width and at the checkpoint's width;
(d) the in-context prompt: rows of the reference context (speaker vector, codecEmbedIcl) through prefill against the oracle's talker
    logits under teacher forcing, and the non-streaming tail (reference codes prepended, proportional cut).
Tolerances: exact-f32 chains: max |err| <= 1e-5 max |ref| per tap, about three times what was observed (values go to profiles/ via `record`)."""
import numpy as np
import pytest
import torch

import mlx_audio_swift_amd as mas
from gpu_util import record
from oracle import ecapa as oe
from oracle import mel as omel
from oracle import mimi_encoder as om
from oracle import qwen3tts as oq
from test_gpu_qwen3tts import _host_cfg

pytestmark = pytest.mark.gpu

TOL = 1e-5            # observed on MI355X (profiles/r03_parity_observed.json): taps <= 3.5e-6, x-vector 6.9e-7, codes agree 100 %


def _audio(n, seed=0, sr=24000.0):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    x = 0.25 * np.sin(2 * np.pi * 180 * t) * (1 + 0.5 * np.sin(2 * np.pi * 2.5 * t)) + 0.08 * np.sin(2 * np.pi * 1330 * t + 1.0)
    return (x + 0.05 * rng.standard_normal(n)).astype(np.float32)


def _spk_host(c: oe.EcapaConfig):
    return mas.qwen3tts.Qwen3TTSSpeakerEncoderConfiguration(**{k: getattr(c, k) for k in mas.qwen3tts.Qwen3TTSSpeakerEncoderConfiguration.__dataclass_fields__})


def _enc_host(c: om.MimiEncoderConfig):
    f = mas.qwen3tts.Qwen3TTSTokenizerEncoderConfiguration.__dataclass_fields__
    return mas.qwen3tts.Qwen3TTSTokenizerEncoderConfiguration(**{k: getattr(c, k) for k in f if hasattr(c, k)})


# speaker encoder whose x-vector is as wide as oq.TINY's talker; tokenizer encoder producing oq.TINY's code groups (4 x 96 codes) at the
# decoder's hop (3 * 2 * 2 = 12 samples per frame); head_dim 16
SPK_SMALL = oe.EcapaConfig(mel_dim=20, enc_dim=256, enc_channels=(32, 32, 32, 32, 96), enc_kernel_sizes=(5, 3, 3, 3, 1), enc_dilations=(1, 2, 3, 4, 1),
                           enc_attention_channels=16, enc_res2net_scale=4, enc_se_channels=8)
ENC_SMALL = om.MimiEncoderConfig(num_filters=8, upsampling_ratios=(3, 2), hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                                 intermediate_size=64, sampling_rate=240, frame_rate=20.0, codebook_dim=16, codebook_size=96, num_quantizers=6,
                                 valid_num_quantizers=4)


def _model(spk=SPK_SMALL, enc=ENC_SMALL, ocfg=oq.TINY):
    W = oq.make_synthetic_weights(ocfg)
    Wd = oq.make_synthetic_decoder_weights(ocfg.decoder)
    hc = _host_cfg(ocfg)
    allw = {("talker." + k): v for k, v in W.items()}
    allw.update(Wd)
    Ws = We = None
    if spk is not None:
        hc.speaker_encoder = _spk_host(spk)
        Ws = oe.make_synthetic_weights(spk)
        allw.update({"speaker_encoder." + k: v for k, v in Ws.items()})
    if enc is not None:
        hc.tokenizer_encoder = _enc_host(enc)
        hc.encoder_valid_num_quantizers = enc.valid_num_quantizers
        We = om.make_synthetic_weights(enc)
        allw.update({"encoder_model." + k: v for k, v in We.items()})
    dev = mas.Qwen3TTSModel.from_weights(hc, allw)
    return dev, oq.Qwen3TTSOracle(ocfg, W), oq.SpeechDecoderOracle(ocfg.decoder, Wd), Ws, We


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(float(np.abs(b).max()), 1e-30))


def test_mel_nfft_1024_matches_oracle():
    for n, seed in ((24000 * 3 + 77, 1), (5000, 2)):
        x = _audio(n, seed)
        got = mas.dsp.compute_mel_spectrogram(x, 24000, 1024, 256, 128)
        ref = omel.compute_mel_spectrogram(x, 24000, 1024, 256, 128)
        assert got.shape == ref.shape
        d = np.abs(got - ref)
        record(f"mel_nfft1024_n{n}", max_abs=float(d.max()), rms=float(np.sqrt(np.mean(d.astype(np.float64) ** 2))), tol_max=6e-5, tol_rms=2e-6)
        assert d.max() < 6e-5 and float(np.sqrt(np.mean(d.astype(np.float64) ** 2))) < 2e-6             # observed 3.0e-5 / 6.9e-7


@pytest.mark.parametrize("which", ["small", "checkpoint"])
def test_speaker_encoder_matches_oracle(which):
    cfg = SPK_SMALL if which == "small" else oe.EcapaConfig()             # the front end does not depend on the talker's width
    dev, _, _, Ws, _ = _model(spk=cfg, enc=None)
    orc = oe.EcapaOracle(cfg, Ws)
    x = _audio(24000 * 2 + 311, 5)
    mel = omel.compute_mel_spectrogram(x, cfg.sample_rate, 1024, 256, cfg.mel_dim)
    ref, inter = orc(mel[None], return_intermediates=True)
    n_blocks = len(cfg.enc_channels)
    names = [f"block{i}" for i in range(n_blocks - 1)] + ["mfa", "asp"]
    worst = 0.0
    for stage, nm in enumerate(names):
        got = dev.reference_tap(0, x, stage)
        r = inter[nm][0]
        assert got.shape == r.shape, (nm, got.shape, r.shape)
        e = _rel(got, r)
        worst = max(worst, e)
        assert e <= TOL, (nm, e)
    xv = dev.extract_speaker_embedding(x)
    e = _rel(xv, ref[0])
    record(f"speaker_encoder_{which}", xvector_max_rel=e, worst_tap_max_rel=worst, tol=TOL)
    assert xv.shape == (cfg.enc_dim,) and e <= TOL, e
    assert np.array_equal(xv, dev.extract_speaker_embedding(x[None]))          # [1, n] input, deterministic
    with pytest.raises(mas.AudioGenerationError):
        dev.extract_speaker_embedding(x[:2000])                                 # fewer than 16 mel frames


def _check_codes(o, audio, got, hidden_dev):
    """every decision of the engine, replayed in the oracle's arithmetic on the engine's own residual path (latent of the tap, minus the
    code vectors the engine picked so far): its pick is the nearest code up to float rounding of the distances"""
    cfg = o.cfg
    ref = o.encode(audio[None, None])[0]
    assert got.shape == ref.shape, (got.shape, ref.shape)
    x = torch.as_tensor(hidden_dev[None])
    agree = float((got == ref).mean())
    checked = 0
    with torch.no_grad():
        for grp, rows in (("rvq_first", [0]), ("rvq_rest", list(range(1, got.shape[0])))):
            p = f"quantizer.{grp}"
            z = torch.nn.functional.conv1d(x, o.w[p + ".input_proj.weight"].permute(0, 2, 1).contiguous())[0].T      # [T, cd]
            resid = z.clone()
            for li, row in enumerate(rows):
                q = f"{p}.vq.layers.{li}.codebook"
                emb = o.w[q + ".embedding_sum"] / torch.clamp(o.w[q + ".cluster_usage"], min=1e-5)[:, None]
                dist = (emb * emb).sum(-1) / 2 - resid @ emb.T
                g = torch.as_tensor(got[row].astype(np.int64))
                best = dist.min(-1).values
                pick = dist.gather(1, g[:, None])[:, 0]
                scale = float(dist.abs().max())
                assert bool(((pick - best) <= 1e-4 * scale).all()), (grp, li, float((pick - best).max()), scale)
                checked += int(z.shape[0])
                resid = resid - emb[g]
    return agree, checked


@pytest.mark.parametrize("which", ["small", "checkpoint"])
def test_tokenizer_encoder_matches_oracle(which):
    if which == "small":
        cfg, n = ENC_SMALL, 4000
    else:                                   # the published encoder: 64 filters, ratios 8 6 5 4, 8 x 512 transformer, 2048 x 256 codebooks
        cfg, n = om.MimiEncoderConfig(), 24000 + 1234
    dev, _, _, _, We = _model(spk=None, enc=cfg)                            # the encoder does not depend on the talker either
    o = om.MimiEncoderOracle(cfg, We)
    x = _audio(n, 9, float(cfg.sampling_rate))
    with torch.no_grad():
        a = torch.as_tensor(x[None, None])
        r0 = o.seanet(a)
        r1 = o.transformer(r0)
        r2 = o.sconv("downsample.conv", r1, 2 * cfg.downsample_stride, stride=cfg.downsample_stride, causal=cfg.use_causal_conv, mode="edge", bias=False)
    errs = []
    hid = None
    for stage, r in enumerate((r0, r1, r2)):
        got = dev.reference_tap(1, x, stage)
        assert got.shape == tuple(r.shape[1:]), (stage, got.shape, r.shape)
        errs.append(_rel(got, r[0].numpy()))
        hid = got
    codes = dev.encode_audio(x)
    agree, checked = _check_codes(o, x, codes, hid)
    record(f"tokenizer_encoder_{which}", seanet_max_rel=errs[0], transformer_max_rel=errs[1], latent_max_rel=errs[2], codes_agree=agree,
           decisions_checked=checked, tol=TOL, frames=int(codes.shape[1]))
    assert max(errs) <= TOL, errs
    assert codes.shape[0] == min(cfg.valid_num_quantizers, cfg.num_quantizers) and agree > 0.9, agree
    # lengths: every stride rounds up (Conv.swift:206-226)
    for m in (1, cfg.downsample_stride * int(np.prod(cfg.upsampling_ratios)) + 1):
        T = -(-m // int(np.prod(cfg.upsampling_ratios)))
        assert dev.encode_audio(x[:m]).shape[1] == -(-T // cfg.downsample_stride)


class _Tok:
    def encode(self, s):
        out, i = [], 0
        special = {"<|im_start|>": 480, "<|im_end|>": 481, "assistant": 482, "user": 483, "\n": 484}
        while i < len(s):
            for k, v in special.items():
                if s.startswith(k, i):
                    out.append(v); i += len(k); break
            else:
                out.append(ord(s[i]) % 400); i += 1
        return out


def test_in_context_prompt_prefill_frames_and_tail():
    dev, orc, odec, Ws, We = _model()
    dev.tokenizer = _Tok()
    cfg = oq.TINY
    x = _audio(4000, 21, 240.0)
    ctx = dev.reference_audio_context(x)
    assert dev.reference_audio_context(x) is ctx                                   # cachedReferenceAudioContext: same array object
    T = ctx.codes.shape[1]
    assert ctx.codes.shape == (4, T) and T == -(-(-(-4000 // 6)) // 2) and ctx.speaker_row == 0 and ctx.first_frame_row == 1
    p = dev.prepare_icl_generation_inputs("Hello there", x, "ref words", "auto")
    V = cfg.talker.vocab_size
    # layout (prepareICLGenerationInputs :753-837)
    ids = _Tok().encode("<|im_start|>assistant\nHello there<|im_end|>\n<|im_start|>assistant\n")
    rid = _Tok().encode("<|im_start|>assistant\nref words<|im_end|>\n")
    body = rid[3:-2] + ids[3:-5] + [cfg.tts_eos_token_id]
    assert p.text_ids.tolist() == ids[:3] + [cfg.tts_pad_token_id] * 4 + [cfg.tts_bos_token_id] + body + [cfg.tts_pad_token_id] * (T + 1)
    assert p.codec_ids.tolist() == [-1] * 3 + [cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id, V + 0, cfg.codec_pad_id] \
        + [cfg.codec_pad_id] * len(body) + [cfg.codec_bos_id] + [V + 1 + i for i in range(T)]
    assert len(p.trailing_ids) == 0
    # the oracle's view of the same prompt: extra rows = [speaker vector, codecEmbedIcl rows]
    xv = oe.EcapaOracle(SPK_SMALL, Ws)(omel.compute_mel_spectrogram(x, 24000, 1024, 256, SPK_SMALL.mel_dim)[None])[0]
    rows = np.concatenate([np.asarray(ctx.speaker_embedding)[None], orc.codec_embed_icl(ctx.codes).numpy()], 0)
    assert _rel(ctx.speaker_embedding, xv) <= TOL
    # teacher-forced frames: the device's own greedy codes forced through the oracle, talker logits compared through the NEXT frame's code
    gp = mas.Qwen3TTSGenerateParameters(max_tokens=6, temperature=0.0, top_k=0, repetition_penalty=1.0)
    codes = dev.generate_codes([p], gp)[0]
    assert len(codes) >= 1
    params = dict(temperature=0.0, top_p=1.0, top_k=0, repetition_penalty=1.0, min_p=0.0, seed=0)
    ref_codes, tl = orc.generate_row(p.text_ids, p.codec_ids, [], params, max_frames=len(codes), forced_codes=codes, extra_rows=rows)
    # greedy agreement wherever the oracle's top-2 margin is not a rounding matter
    for f in range(len(codes)):
        lg = tl[f].copy()
        lg[V - 1024:V] = -np.inf
        lg[cfg.codec_eos_token_id] = tl[f][cfg.codec_eos_token_id]
        top = np.sort(lg)[-2:]
        if top[1] - top[0] > 0.05 * max(1.0, abs(top[1])):
            assert int(np.argmax(lg)) == int(codes[f][0]), f
    # non-streaming tail (:547-563): [reference codes | generated] decoded, the proportional head cut off
    pcm, gen = dev.generate_batch([p], gp, return_codes=True)
    assert np.array_equal(gen[0], codes)
    allc = np.concatenate([ctx.codes.T, codes], 0)                                 # [R + n, G]
    up = dev.samples_per_frame
    wav = np.asarray(odec.decode(allc.T[None]))[0]
    valid = int((allc[:, 0] > 0).sum()) * up
    full = wav[:valid] if 0 < valid < len(wav) else wav
    cut = int(float(T) / float(len(allc)) * float(len(full)))
    want = full[cut:] if 0 < cut < len(full) else full
    assert len(pcm[0]) == len(want), (len(pcm[0]), len(want))
    e = float(np.abs(pcm[0] - want).max() / np.abs(wav).max())
    record("icl_tail_waveform", max_rel=e, tol=5e-5, ref_frames=T, generated=len(codes))
    assert e <= 5e-5, e                                                          # observed 1.2e-5
    # a second, plain row in the same batch is untouched by the reference; rows stay independent
    q = dev.prepare_generation_inputs("Hello there", "auto", None)
    both, bc = dev.generate_batch([p, q], gp, return_codes=True)
    solo, sc = dev.generate_batch([q], gp, return_codes=True)
    assert np.array_equal(bc[0], codes) and np.array_equal(bc[1], sc[0]) and np.array_equal(both[0], pcm[0]) and np.array_equal(both[1], solo[0])
    # streaming decodes only what it generated (:535-546)
    chunks = []
    dev.generate_batch([p], gp, streaming_interval=0.16, on_audio=lambda row, a: chunks.append(a))
    assert sum(len(c) for c in chunks) == len(codes) * up
    # caller-supplied conditioning (Qwen3TTSReferenceConditioning): precomputed codes, no speaker vector -> no speaker position
    ctx2 = dev.add_reference(ctx.codes[:, :5])
    assert ctx2.speaker_row == -1 and ctx2.first_frame_row == 1 + T
    p2 = dev.prepare_icl_generation_inputs("Hello there", conditioning=(ctx2, rid[3:-2], None))
    assert len(p2.codec_ids) == len(p.codec_ids) - 1 - (T - 5) and p2.codec_ids.tolist()[-5:] == [V + 1 + T + i for i in range(5)]
    assert len(dev.generate_codes([p2], gp)[0]) >= 1
    dev.clear_references()
    with pytest.raises(mas.AudioGenerationError):
        dev.generate_codes([p], gp)                                               # its rows are gone


def test_in_context_rows_sharded_over_two_replicas_equal_the_single_handle():
    """mis_qwen3tts_group_generate with in-context prompts: the reference contexts of the first model are registered on every replica in
    the same order (Qwen3TTSModel._sync_references), each replica decodes [reference codes | generated] for its own rows.  Two replicas
    on one GPU (same weights), three rows: in-context, plain, in-context with a second recording."""
    dev, orc, odec, Ws, We = _model()
    dev2, *_ = _model()
    dev.tokenizer = _Tok()
    xa, xb = _audio(4000, 21, 240.0), _audio(4100, 22, 240.0)
    pa = dev.prepare_icl_generation_inputs("Hello there", xa, "ref words", "auto")
    pq = dev.prepare_generation_inputs("Hello there", "auto", None)
    pb = dev.prepare_icl_generation_inputs("Other words", xb, "second ref", "auto")
    assert pb.reference is not pa.reference and pb.reference.first_frame_row > pa.reference.first_frame_row
    gp = mas.Qwen3TTSGenerateParameters(max_tokens=5, temperature=0.8, top_k=20, seed=5)
    one, c1 = dev.generate_batch([pa, pq, pb], gp, return_codes=True)
    two, c2 = dev.generate_batch([pa, pq, pb], gp, return_codes=True, replicas=[dev, dev2])
    assert len(dev2._ref_log) == len(dev._ref_log) == 2
    for r in range(3):
        assert np.array_equal(c1[r], c2[r]) and np.array_equal(one[r], two[r]), r
    up = dev.samples_per_frame
    assert len(one[1]) == len(c1[1]) * up or len(c1[1]) == 0       # the plain row is decoded on its own codes only


class _WordTok:
    """the reference fixture's tokenizer (Tests/MLXAudioTTSTests.swift:546-612): lower-cased whitespace words over an 18-entry vocabulary"""
    vocab = {"<bos>": 0, "<pad>": 1, "<eos>": 2, "<unk>": 3, "<|im_start|>": 4, "<|im_end|>": 5, "assistant": 6, "user": 7, "one": 8, "two": 9,
             "three": 10, "four": 11, "five": 12, "target": 13, "voice": 14, "prompt": 15, "sample": 16, "english": 17}

    def encode(self, s):
        import re
        return [self.vocab.get(w, 3) for w in re.findall(r"<\|im_start\|>|<\|im_end\|>|[a-z0-9]+", s.lower())]


def _reference_fixture(tts_model_type, include_speech_encoder, spk_id=None):
    """makeTinyQwen3TTSModel (Tests/MLXAudioTTSTests.swift:615-687) at widths the engine takes (hidden 256 instead of 16): 3072 / 2048
    vocabularies, two code groups, encoder_valid_num_quantizers 2, the default Mimi encoder config when `encoder_config` is present"""
    import dataclasses
    ocfg = oq.Qwen3TTSConfig(
        talker=dataclasses.replace(oq.TINY.talker, vocab_size=3072), predictor=dataclasses.replace(oq.TINY.predictor, vocab_size=2048, num_hidden_layers=1),
        num_code_groups=2, text_hidden_size=128, text_vocab_size=64, codec_eos_token_id=3050, codec_think_id=3051, codec_nothink_id=3052,
        codec_think_bos_id=3053, codec_think_eos_id=3054, codec_pad_id=3055, codec_bos_id=3056, tts_pad_token_id=21, tts_bos_token_id=22,
        tts_eos_token_id=23, decoder=dataclasses.replace(oq.TINY.decoder, codebook_size=2048, num_quantizers=2))
    hc = _host_cfg(ocfg)
    hc.codec_language_id = {"english": 3057}
    hc.tts_model_type = tts_model_type
    hc.spk_id = spk_id
    W = oq.make_synthetic_weights(ocfg)
    allw = {("talker." + k): v for k, v in W.items()}
    allw.update(oq.make_synthetic_decoder_weights(ocfg.decoder))
    if include_speech_encoder:
        enc = om.MimiEncoderConfig(valid_num_quantizers=2)
        hc.tokenizer_encoder = _enc_host(enc)
        hc.encoder_valid_num_quantizers = 2
        allw.update({"encoder_model." + k: v for k, v in om.make_synthetic_weights(enc).items()})
    dev = mas.Qwen3TTSModel.from_weights(hc, allw)
    dev.tokenizer = _WordTok()
    return dev


def test_reference_conditioning_pins_of_the_reference_test_suite():
    """The reference's own Qwen3TTS suite (Tests/MLXAudioTTSTests.swift:936-1084) pins shapes and control flow, not numbers; the same pins
    here: prepareReferenceConditioning on a voice_design model with a speech encoder (no speaker vector, codes [2, T > 0], reference text
    ids, "english" -> codec language id 3057), generate from the prepared conditioning and through the raw refAudio / refText arguments,
    direct and streaming (tokens > 0, one info, audio 1-D), and a custom_voice model without an encoder refusing reference conditioning."""
    import wave
    import os
    with wave.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "intention.wav"), "rb") as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16).astype(np.float32) / 32768.0
    ref_audio = np.ascontiguousarray(pcm[:24000])                                   # loadTTSNetworkFixture(sampleRate: 24_000, maxSamples: 24_000)
    dev = _reference_fixture("voice_design", True)
    ctx, ref_ids, lang_id = dev.prepare_reference_conditioning(ref_audio, "one two three four five one two three four five", "English")
    assert ctx.speaker_embedding is None and ctx.speaker_row == -1                  # #expect(conditioning.speakerEmbedding == nil)
    assert ctx.codes.ndim == 2 and ctx.codes.shape[0] == 2 and ctx.codes.shape[1] > 0
    assert ctx.codes.shape[1] == -(-(-(-24000 // 960)) // 2)                        # 12.5 Hz: 13 frames of one second
    assert len(ref_ids) > 0 and lang_id == 3057                                     # resolvedLanguage "english" -> codecLanguageID 3057
    gp = mas.Qwen3TTSGenerateParameters(max_tokens=2, temperature=0.7, top_p=0.95, repetition_penalty=1.0)
    text = "target voice prompt one two three four five"
    p = dev.prepare_icl_generation_inputs(text, conditioning=(ctx, ref_ids, lang_id))
    assert p.codec_ids.tolist()[3:8] == [3051, 3053, 3057, 3054, 3055]               # think prefix with the language id, no speaker position
    audio = dev.generate_batch([p], gp)[0]
    assert audio.ndim == 1 and audio.shape[0] > 0
    ev = list(dev.generate_stream_batch([p], gp, streaming_interval=0.05))
    assert sum(isinstance(e, mas.TokenEvent) for e in ev) > 0 and sum(isinstance(e, mas.InfoEvent) for e in ev) == 1
    last = [e for e in ev if isinstance(e, mas.AudioEvent)][-1]
    assert last.audio.ndim == 1
    raw = dev.generate(text, voice=None, ref_audio=ref_audio, ref_text="one two three four five one two three four five", language="English",
                       generation_parameters=gp)
    assert raw.ndim == 1 and raw.shape[0] > 0
    ev2 = list(dev.generate_stream(text, None, "English", gp, 0.05, ref_audio=ref_audio, ref_text="one two three four five one two three four five"))
    assert sum(isinstance(e, mas.TokenEvent) for e in ev2) > 0 and sum(isinstance(e, mas.InfoEvent) for e in ev2) == 1
    # customVoiceRemainsSeparateFromReferenceConditioning (:1050-1084)
    cv = _reference_fixture("custom_voice", False, spk_id={"ryan": 100})
    out = cv.generate(text, voice="ryan", language="English", generation_parameters=gp)
    assert out.ndim == 1 and out.shape[0] > 0
    with pytest.raises(mas.AudioGenerationError) as e:
        cv.prepare_reference_conditioning(ref_audio, "one two three four five one two three four five", "English")
    assert e.value.case == "invalidInput"
