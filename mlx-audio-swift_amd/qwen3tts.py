"""Host mirror of Qwen3TTSModel : SpeechGenerationModel (Sources/MLXAudioTTS/Models/Qwen3TTS/Qwen3TTS.swift:12-133).

Device work (talker + code-predictor frame loop, sampleToken, speech-tokenizer decoder) is behind mis_qwen3tts_* in
libmi_speech.so.  Host logic mirrored here: the ChatML prompt template and the (text id, codec id) layout of
prepareGenerationInputs (Qwen3TTS.swift:883-1000), the per-utterance frame cap (:383) and the stream contract."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .codecs import _tensor_args
from .generation import (AudioEvent, AudioGenerationError, InfoEvent, TokenEvent, check, decode_audio_event,  # noqa: F401
                         stream_events)
from .tts import LlamaTTSConfiguration


def _lm(hidden, layers, ff, heads, kv, hd, vocab):
    return LlamaTTSConfiguration(hidden_size=hidden, num_hidden_layers=layers, intermediate_size=ff, num_attention_heads=heads,
                                 num_key_value_heads=kv, head_dim=hd, vocab_size=vocab, rms_norm_eps=1e-6, rope_theta=1e6,
                                 rope_scaling=None, tie_word_embeddings=False, qk_norm=True, rope_plain=True, rope_ops_in_dtype=True)


@dataclass
class Qwen3TTSDecoderConfiguration:
    """Qwen3TTSTokenizerDecoderConfig (Qwen3TTSConfig.swift:307-385)"""
    latent_dim: int = 1024
    codebook_dim: int = 512
    codebook_size: int = 2048
    decoder_dim: int = 1536
    hidden_size: int = 512
    intermediate_size: int = 1024
    head_dim: int = 64
    num_attention_heads: int = 16
    num_hidden_layers: int = 8
    num_key_value_heads: int = 16
    num_quantizers: int = 16
    num_semantic_quantizers: int = 1
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    upsample_rates: tuple = (8, 5, 4, 3)
    upsampling_ratios: tuple = (2, 2)


@dataclass
class Qwen3TTSConfiguration:
    """Qwen3TTSModelConfig + Qwen3TTSTalkerConfig (Qwen3TTSConfig.swift:200-305,532-583)"""
    talker: LlamaTTSConfiguration = field(default_factory=lambda: _lm(1024, 28, 3072, 16, 8, 128, 3072))
    predictor: LlamaTTSConfiguration = field(default_factory=lambda: _lm(1024, 5, 3072, 16, 8, 128, 2048))
    num_code_groups: int = 16
    text_hidden_size: int = 2048
    text_vocab_size: int = 151936
    codec_eos_token_id: int = 2150
    codec_think_id: int = 2154
    codec_nothink_id: int = 2155
    codec_think_bos_id: int = 2156
    codec_think_eos_id: int = 2157
    codec_pad_id: int = 2148
    codec_bos_id: int = 2149
    codec_language_id: dict | None = None
    spk_id: dict | None = None              # CustomVoice: speaker name -> codec token id (int or [int, ...]: first), Qwen3TTSConfig.swift:118-147
    spk_is_dialect: dict | None = None      # speaker name -> false | dialect name (a codec_language_id key), :153-196
    tts_model_type: str = "base"            # "custom_voice": `voice` = "speaker[, instruction]" (Qwen3TTS.swift:361-371)
    tts_pad_token_id: int = 151671
    tts_bos_token_id: int = 151672
    tts_eos_token_id: int = 151673
    sample_rate: int = 24000
    decoder: Qwen3TTSDecoderConfiguration = field(default_factory=Qwen3TTSDecoderConfiguration)

    @classmethod
    def from_dict(cls, d: dict, tokenizer: dict | None = None) -> "Qwen3TTSConfiguration":
        """Qwen3TTSModelConfig (config.json: talker_config { code_predictor_config }, tts_*_token_id, sample_rate) plus the speech
        tokenizer's decoder_config (speech_tokenizer/config.json); defaults as Qwen3TTSConfig.swift:47-63,267-295,362-385,582-587."""
        t = d.get("talker_config") or {}
        cp = t.get("code_predictor_config") or {}

        def lm(c, layers, vocab):
            return _lm(c.get("hidden_size", 1024), c.get("num_hidden_layers", layers), c.get("intermediate_size", 3072),
                       c.get("num_attention_heads", 16), c.get("num_key_value_heads", 8), c.get("head_dim", 128), c.get("vocab_size", vocab))
        talker, pred = lm(t, 28, 3072), lm(cp, 5, 2048)
        talker.rms_norm_eps = t.get("rms_norm_eps", 1e-6); talker.rope_theta = t.get("rope_theta", 1e6)
        pred.rms_norm_eps = cp.get("rms_norm_eps", 1e-6); pred.rope_theta = cp.get("rope_theta", 1e6)
        dc = (tokenizer or {}).get("decoder_config") or {}
        dec = Qwen3TTSDecoderConfiguration(**{k: (tuple(dc[k]) if isinstance(dc[k], list) else dc[k])
                                              for k in Qwen3TTSDecoderConfiguration.__dataclass_fields__ if k in dc})
        return cls(talker=talker, predictor=pred, num_code_groups=t.get("num_code_groups", 16),
                   text_hidden_size=t.get("text_hidden_size", 2048), text_vocab_size=t.get("text_vocab_size", 151936),
                   codec_eos_token_id=t.get("codec_eos_token_id", 2150), codec_think_id=t.get("codec_think_id", 2154),
                   codec_nothink_id=t.get("codec_nothink_id", 2155), codec_think_bos_id=t.get("codec_think_bos_id", 2156),
                   codec_think_eos_id=t.get("codec_think_eos_id", 2157), codec_pad_id=t.get("codec_pad_id", 2148),
                   codec_bos_id=t.get("codec_bos_id", 2149), codec_language_id=t.get("codec_language_id"),
                   spk_id=t.get("spk_id"), spk_is_dialect=t.get("spk_is_dialect"), tts_model_type=d.get("tts_model_type", "base"),
                   tts_pad_token_id=d.get("tts_pad_token_id", 151671), tts_bos_token_id=d.get("tts_bos_token_id", 151672),
                   tts_eos_token_id=d.get("tts_eos_token_id", 151673), sample_rate=d.get("sample_rate", 24000), decoder=dec)

    def to_c(self) -> "_lib.Qwen3TTSConfigC":
        d = self.decoder
        ur = (C.c_int32 * 8)(*d.upsample_rates)
        up = (C.c_int32 * 8)(*d.upsampling_ratios)
        return _lib.Qwen3TTSConfigC(self.talker.to_c(), self.predictor.to_c(), self.num_code_groups, self.text_hidden_size,
                                    self.text_vocab_size, self.codec_eos_token_id, self.tts_pad_token_id, d.latent_dim,
                                    d.codebook_dim, d.codebook_size, d.decoder_dim, d.hidden_size, d.intermediate_size, d.head_dim,
                                    d.num_attention_heads, d.num_hidden_layers, d.num_key_value_heads, d.num_quantizers,
                                    d.num_semantic_quantizers, d.rms_norm_eps, d.rope_theta, len(d.upsample_rates), ur,
                                    len(d.upsampling_ratios), up, self.sample_rate)


@dataclass
class Qwen3TTSGenerateParameters:
    """defaultGenerationParameters / resolveVoiceDesignGenerationSettings (Qwen3TTS.swift:30-37,651-664)"""
    max_tokens: int = 4096
    temperature: float = 0.9
    top_p: float = 1.0
    top_k: int = 0
    repetition_penalty: float = 1.05
    min_p: float = 0.0
    seed: int = 0
    row_offset: int = 0

    def to_c(self):
        return _lib.Qwen3TTSParamsC(int(self.max_tokens), float(self.temperature), float(self.top_p), int(self.top_k),
                                    float(self.repetition_penalty), float(self.min_p), int(self.seed), int(self.row_offset))


def read_safetensors(path: str) -> dict:
    """name -> torch tensor (bf16 / f16 / f32 as stored) or numpy uint32 array (quantised words).  MLX writes U32 for packed weights."""
    import json
    import torch
    out = {}
    with open(path, "rb") as f:
        n = int.from_bytes(f.read(8), "little")
        hdr = json.loads(f.read(n))
        blob = f.read()
    kinds = {"BF16": torch.bfloat16, "F16": torch.float16, "F32": torch.float32}
    for k, e in hdr.items():
        if k == "__metadata__":
            continue
        a, b = e["data_offsets"]
        raw = np.frombuffer(blob, np.uint8, b - a, a)
        if e["dtype"] in kinds:
            out[k] = torch.frombuffer(bytearray(raw.tobytes()), dtype=kinds[e["dtype"]]).reshape(e["shape"]) if b > a else torch.zeros(e["shape"], dtype=kinds[e["dtype"]])
        elif e["dtype"] in ("U32", "I32"):
            out[k] = raw.view(np.uint32).reshape(e["shape"]).copy()
        elif e["dtype"] in ("I64", "BOOL", "U8"):
            continue                                                     # bookkeeping tensors (codebook `initialized`, ...)
        else:
            raise AudioGenerationError(3, f"unsupported safetensors dtype {e['dtype']} for {k}")
    return out


def _mlx_conv_shape(shape) -> bool:
    """checkArrayShapeQwen3 (Qwen3TTSSpeechTokenizer.swift:1445-1455): does a 3-D conv weight already look like MLX's [out, k, in]?"""
    if len(shape) != 3:
        return False
    _, d2, d3 = shape
    if d2 == 1:
        return d3 > 64
    if d3 == 1:
        return d2 <= 64
    return d2 < d3


def sanitize_speech_tokenizer(weights: dict) -> dict:
    """Qwen3TTSSpeechTokenizer.sanitize (:1093-1440), decoder side: strip the speech_tokenizer./decoder_model. prefixes, keep the
    decoder codebooks' cluster_usage / embedding_sum (`_codebook.` -> `.codebook.`), transpose PyTorch conv weights to MLX's layout
    when the shape heuristic says they are not already (transposed convs [in, out, k] -> [out, k, in], convs [out, in, k] ->
    [out, k, in]), rename upsample.X.Y -> upsample.X.layers.Y.  Encoder / speaker-encoder keys are dropped (not built)."""
    import re
    out = {}
    for raw, v in weights.items():
        k = raw
        stripped = True
        while stripped:
            stripped = False
            for pre in ("speech_tokenizer.", "encoder_model.", "decoder_model."):
                if k.startswith(pre):
                    k = k[len(pre):]; stripped = True; break
        if not k or k.startswith("encoder.") or "speaker_encoder" in k or "initialized" in k:
            continue
        if "_codebook.cluster_usage" in k or "_codebook.embedding_sum" in k:
            base, leaf = k.rsplit("._codebook.", 1)
            out[f"{base}.codebook.{leaf}"] = v
            continue
        shape = tuple(v.shape)
        is_tconv = ("upsample" in k and ".0.conv.weight" in k) or ("decoder.decoder" in k and "block.1.conv.weight" in k)
        if len(shape) == 3 and not _mlx_conv_shape(shape):
            if is_tconv:
                v = v.permute(1, 2, 0).contiguous()
            elif "conv.weight" in k or "_proj.weight" in k:
                v = v.permute(0, 2, 1).contiguous()
        k = re.sub(r"upsample\.(\d+)\.(\d+)", r"upsample.\1.layers.\2", k)
        out[k] = v
    return out


@dataclass
class PreparedPrompt:
    """One utterance as prepareGenerationInputs lays it out: prefill positions as (text id | -1, codec id | -1) pairs, the
    trailing text ids added to the generated frames, and the number of text tokens (frame cap, Qwen3TTS.swift:381-383)."""
    text_ids: np.ndarray
    codec_ids: np.ndarray
    trailing_ids: np.ndarray
    target_token_count: int = 0


class Qwen3TTSModel:
    def __init__(self, config: Qwen3TTSConfiguration, device: int = 0):
        self.configuration = config
        self.device = device
        self.tokenizer = None
        self._h = C.c_void_p()
        cc = config.to_c()
        check(_lib.lib().mis_qwen3tts_create(C.byref(cc), device, C.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().mis_qwen3tts_destroy(h)

    @classmethod
    def from_weights(cls, config, weights: dict, device: int = 0) -> "Qwen3TTSModel":
        m = cls(config, device)
        for k, v in weights.items():
            m.set_tensor(k, v)
        m.finalize()
        return m

    # -- fromModelDirectory / fromPretrained (Qwen3TTS.swift:1122-1275) -----------------------------------------------------------
    @classmethod
    def from_model_directory(cls, model_dir: str, device: int = 0) -> "Qwen3TTSModel":
        """config.json + *.safetensors (talker, `talker.` prefix stripped by sanitize :357-365; quantised paths = those with a
        `.scales` companion, group size / bits from `quantization` and its per-layer overrides :1157-1170) + speech_tokenizer/
        (config.json, *.safetensors through `sanitize_speech_tokenizer`).  Speaker-encoder / tokenizer-encoder tensors are skipped
        (voice cloning is not built)."""
        import json
        import os
        with open(os.path.join(model_dir, "config.json")) as f:
            cj = json.load(f)
        st_dir = os.path.join(model_dir, "speech_tokenizer")
        tj = {}
        if os.path.isdir(st_dir) and os.path.exists(os.path.join(st_dir, "config.json")):
            with open(os.path.join(st_dir, "config.json")) as f:
                tj = json.load(f)
        m = cls(Qwen3TTSConfiguration.from_dict(cj, tj), device)
        weights = {}
        for fn in sorted(os.listdir(model_dir)):
            if fn.endswith(".safetensors"):
                weights.update(read_safetensors(os.path.join(model_dir, fn)))
        talker = {k[len("talker."):]: v for k, v in weights.items() if k.startswith("talker.")}
        quant = cj.get("quantization") or cj.get("quantization_config") or {}
        for name, arr in talker.items():
            if name.endswith(".scales") or name.endswith(".biases"):
                continue
            base = name[: -len(".weight")] if name.endswith(".weight") else name
            if base + ".scales" in talker:
                per = quant.get("talker." + base) or quant.get(base) or {}
                gs, bits = int(per.get("group_size", quant.get("group_size", 64))), int(per.get("bits", quant.get("bits", 4)))
                m.set_quantized_tensor(name, np.asarray(arr).view(np.uint32), talker[base + ".scales"], talker[base + ".biases"], gs, bits)
            else:
                m.set_tensor(name, arr)
        if not os.path.isdir(st_dir):
            raise AudioGenerationError(1, "speech_tokenizer directory not found: speech decoding unavailable")
        tw = {}
        for fn in sorted(os.listdir(st_dir)):
            if fn.endswith(".safetensors"):
                tw.update(read_safetensors(os.path.join(st_dir, fn)))
        for name, arr in sanitize_speech_tokenizer(tw).items():
            if name.startswith("decoder."):
                m.set_tensor(name, arr)
        m.finalize()
        return m

    @classmethod
    def from_pretrained(cls, model_repo: str, device: int = 0) -> "Qwen3TTSModel":
        import os
        if os.path.isdir(model_repo):
            return cls.from_model_directory(model_repo, device)
        raise AudioGenerationError(1, f"model repo {model_repo!r} is not a local directory (no network access)")

    def set_tensor(self, name: str, arr):
        keep, ptr, dt, shape = _tensor_args(arr)
        sh = (C.c_int64 * len(shape))(*shape)
        check(_lib.lib().mis_qwen3tts_set_tensor(self._h, name.encode(), ptr, dt, sh, len(shape)))

    def set_quantized_tensor(self, name: str, wq, scales, biases, group_size: int = 64, bits: int = 8):
        """A tensor of a quantised checkpoint: wq uint32 [N, K*bits/32], scales / biases [N, K/group_size]."""
        wq = np.ascontiguousarray(wq, dtype=np.uint32)
        ks, ps, ds, ss = _tensor_args(scales)
        kb, pb, db, sb = _tensor_args(biases)
        if ds != db or tuple(ss) != tuple(sb):
            raise AudioGenerationError(3, "scales and biases must share dtype and shape")
        N, K = int(ss[0]), int(ss[1]) * group_size
        check(_lib.lib().mis_qwen3tts_set_tensor_quantized(self._h, name.encode(), wq.ctypes.data, ps, pb, ds, N, K, group_size, bits))

    def finalize(self):
        check(_lib.lib().mis_qwen3tts_finalize(self._h))

    # -- protocol surface ------------------------------------------------------------------------------
    @property
    def sample_rate(self) -> int:
        return self.configuration.sample_rate

    @property
    def default_generation_parameters(self) -> Qwen3TTSGenerateParameters:
        return Qwen3TTSGenerateParameters()

    @property
    def samples_per_frame(self) -> int:
        return int(_lib.lib().mis_qwen3tts_samples_per_frame(self._h))

    @staticmethod
    def parse_custom_voice_prompt(voice: str | None):
        """parseCustomVoicePrompt (Qwen3TTS.swift:571-594): "speaker[, instruction]" -> (speaker, instruction | None); None if empty."""
        v = (voice or "").strip()
        if not v:
            return None
        if "," not in v:
            return v, None
        speaker, instruction = v.split(",", 1)
        speaker, instruction = speaker.strip(), instruction.strip()
        if not speaker:
            return v, None
        return speaker, (instruction or None)

    def prepare_generation_inputs(self, text: str, language: str = "auto", instruct: str | None = None,
                                  speaker: str | None = None) -> PreparedPrompt:
        """prepareGenerationInputs (Qwen3TTS.swift:883-1000).  The CustomVoice speaker (:914-936,957-962) is the talker's input
        embedding of the speaker's codec token, spliced between the think prefix and (pad, bos): on this side one more codec id, the
        engine embeds it with the same table; a dialect speaker overrides the language id."""
        if self.tokenizer is None:
            raise AudioGenerationError(1, "Qwen3TTS requires the text tokenizer to be loaded")
        cfg = self.configuration
        ids = list(self.tokenizer.encode(f"<|im_start|>assistant\n{text}<|im_end|>\n<|im_start|>assistant\n"))
        lang = None
        if language.lower() != "auto" and cfg.codec_language_id:
            lang = cfg.codec_language_id.get(language.lower())
        spk_token = None
        if speaker:
            v = (cfg.spk_id or {}).get(speaker.lower())
            if v is not None:                                           # SpkIdValue.intValue: the int, or the first of a list (0 if empty)
                spk_token = (int(v[0]) if v else 0) if isinstance(v, (list, tuple)) else int(v)
            dv = (cfg.spk_is_dialect or {}).get(speaker.lower())
            if isinstance(dv, str) and cfg.codec_language_id and dv in cfg.codec_language_id:
                lang = cfg.codec_language_id[dv]                          # dialect override (:927-935)
        prefix = ([cfg.codec_think_id, cfg.codec_think_bos_id, lang, cfg.codec_think_eos_id] if lang is not None
                  else [cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id])
        codec = prefix + ([spk_token] if spk_token is not None else []) + [cfg.codec_pad_id, cfg.codec_bos_id]
        t, c = [], []
        if instruct:
            ins = list(self.tokenizer.encode(f"<|im_start|>user\n{instruct}<|im_end|>\n"))
            t += ins; c += [-1] * len(ins)
        t += ids[:3]; c += [-1, -1, -1]                                  # role: <|im_start|>assistant\n
        pad_count = len(codec) - 2
        t += [cfg.tts_pad_token_id] * pad_count + [cfg.tts_bos_token_id]  # (pad..., bos) + codec[:-1]
        c += codec[:-1]
        t += [ids[3]]; c += [codec[-1]]                                   # first text token + codec_bos
        trailing = ids[4:len(ids) - 5] + [cfg.tts_eos_token_id]
        return PreparedPrompt(np.asarray(t, np.int32), np.asarray(c, np.int32), np.asarray(trailing, np.int32),
                              len(self.tokenizer.encode(text)))

    def _marshal(self, prompts):
        B = len(prompts)
        if B == 0:
            raise AudioGenerationError(3, "empty batch")
        P = max(len(p.text_ids) for p in prompts)
        Tt = max(max(len(p.trailing_ids) for p in prompts), 1)
        t = np.full((B, P), -1, np.int32); c = np.full((B, P), -1, np.int32); tr = np.zeros((B, Tt), np.int32)
        pl = np.zeros(B, np.int32); tl = np.zeros(B, np.int32)
        for b, p in enumerate(prompts):
            n = len(p.text_ids)
            t[b, :n] = p.text_ids; c[b, :n] = p.codec_ids; pl[b] = n
            tl[b] = len(p.trailing_ids); tr[b, :tl[b]] = p.trailing_ids
        return t, c, pl, P, tr, tl, Tt

    def _row_caps(self, prompts, gp):
        caps = np.asarray([min(gp.max_tokens, max(75, p.target_token_count * 6)) if p.target_token_count > 0 else gp.max_tokens
                           for p in prompts], np.int32)                  # effectiveMaxTokens (:383)
        return caps

    def generate_codes(self, prompts, generation_parameters: Qwen3TTSGenerateParameters | None = None):
        """Frame loop only: list of [n_frames, num_code_groups] int32 arrays."""
        gp = generation_parameters or self.default_generation_parameters
        t, c, pl, P, tr, tl, Tt = self._marshal(prompts)
        B = len(prompts)
        caps = self._row_caps(prompts, gp)
        gpc = gp.to_c()
        gpc.max_frames = int(caps.max())
        out = C.c_void_p(); stride = C.c_int64(); nf = (C.c_int32 * B)()
        check(_lib.lib().mis_qwen3tts_generate_codes(self._h, t.ctypes.data, c.ctypes.data, pl.ctypes.data, P, tr.ctypes.data,
                                                     tl.ctypes.data, Tt, B, C.byref(gpc), caps.ctypes.data, C.byref(out),
                                                     C.byref(stride), nf))
        try:
            G = self.configuration.num_code_groups
            arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(B, max(stride.value, 1), G))
            return [arr[b, : nf[b]].copy() for b in range(B)]
        finally:
            _lib.lib().mis_free(out)

    def decode_codes(self, codes) -> np.ndarray:
        """speechTokenizer.decoder over whole sequences: codes [B, num_quantizers, T] -> [B, T * samples_per_frame]."""
        cd = np.ascontiguousarray(codes, dtype=np.int32)
        B, nq, T = cd.shape
        out = np.zeros((B, T * self.samples_per_frame), np.float32)
        check(_lib.lib().mis_qwen3tts_decode(self._h, cd.ctypes.data, B, T, out.ctypes.data))
        return out

    def decoder_tap(self, codes, stage: int) -> np.ndarray:
        cd = np.ascontiguousarray(codes, dtype=np.int32)
        B, nq, T = cd.shape
        d = self.configuration.decoder
        cap = B * max(d.decoder_dim, d.latent_dim, d.codebook_dim) * T * self.samples_per_frame
        buf = np.zeros(cap, np.float32)
        ch = C.c_int32(); ln = C.c_int64()
        check(_lib.lib().mis_qwen3tts_decoder_tap(self._h, cd.ctypes.data, B, T, stage, buf.ctypes.data, cap, C.byref(ch), C.byref(ln)))
        return buf[: B * ch.value * ln.value].reshape(B, ch.value, ln.value).copy()

    def generate_batch(self, prompts, generation_parameters: Qwen3TTSGenerateParameters | None = None, return_codes: bool = False,
                       streaming_interval: float | None = None, on_audio=None, replicas=None):
        """generateVoiceDesign for a batch of prepared prompts: list of 1-D float32 PCM arrays.  With `on_audio` the decoded
        audio is also delivered in chunks of streaming_interval * 12.5 frames (Qwen3TTS.swift:394-395)."""
        gp = generation_parameters or self.default_generation_parameters
        t, c, pl, P, tr, tl, Tt = self._marshal(prompts)
        B = len(prompts)
        caps = self._row_caps(prompts, gp)
        gpc = gp.to_c()
        gpc.max_frames = int(caps.max())
        pcm = C.c_void_p(); stride = C.c_int64(); plens = (C.c_int64 * B)()
        codes = C.c_void_p(); cstride = C.c_int64(); nf = (C.c_int32 * B)()
        chunk = max(1, int((streaming_interval or 2.0) * 12.5))
        keep = []

        def cb(user, row, kind, payload, n):
            if kind == _lib.EVENT_AUDIO and on_audio is not None:
                on_audio(row, np.ctypeslib.as_array(C.cast(payload, C.POINTER(C.c_float)), shape=(n,)).copy())
        cbc = _lib.EVENT_CB(cb) if on_audio is not None else C.cast(None, _lib.EVENT_CB)
        keep.append(cbc)
        if replicas:          # Qwen3TTSModel objects with the same weights, one per GPU: rows sharded inside the library, each replica
            hs = (C.c_void_p * len(replicas))(*[r._h for r in replicas])          # streams its own rows' chunks (global row indices)
            check(_lib.lib().mis_qwen3tts_group_generate(hs, len(replicas), t.ctypes.data, c.ctypes.data, pl.ctypes.data, P, tr.ctypes.data,
                                                         tl.ctypes.data, Tt, B, C.byref(gpc), caps.ctypes.data, C.byref(pcm), C.byref(stride),
                                                         plens, C.byref(codes), C.byref(cstride), nf, chunk, cbc, None, None))
        else:
            check(_lib.lib().mis_qwen3tts_generate(self._h, t.ctypes.data, c.ctypes.data, pl.ctypes.data, P, tr.ctypes.data, tl.ctypes.data,
                                                   Tt, B, C.byref(gpc), caps.ctypes.data, C.byref(pcm), C.byref(stride), plens,
                                                   C.byref(codes), C.byref(cstride), nf, chunk, cbc, None, None))
        try:
            arr = np.ctypeslib.as_array(C.cast(pcm, C.POINTER(C.c_float)), shape=(B, max(stride.value, 1)))
            out = [arr[b, : plens[b]].copy() for b in range(B)]
            G = self.configuration.num_code_groups
            ca = np.ctypeslib.as_array(C.cast(codes, C.POINTER(C.c_int32)), shape=(B, max(cstride.value, 1), G))
            cl = [ca[b, : nf[b]].copy() for b in range(B)]
        finally:
            _lib.lib().mis_free(pcm)
            _lib.lib().mis_free(codes)
        return (out, cl) if return_codes else out

    def generate(self, text: str, voice: str | None = None, ref_audio=None, ref_text=None, language: str | None = None,
                 generation_parameters: Qwen3TTSGenerateParameters | None = None) -> np.ndarray:
        """generate(text:voice:refAudio:refText:language:generationParameters:) (Qwen3TTS.swift:60-82); `voice` is the
        VoiceDesign instruction."""
        if ref_audio is not None:
            raise AudioGenerationError(5, "in-context voice cloning needs the speech-tokenizer encoder (not built)")
        p = self._prepare(text, voice, language)
        out = self.generate_batch([p], generation_parameters)[0]
        return out if len(out) else np.zeros(1, np.float32)              # generatedCodes.isEmpty -> zeros([1]) (:520-522)

    def _prepare(self, text, voice, language):
        """the non-cloning branch of generate (Qwen3TTS.swift:361-371): CustomVoice models read `voice` as "speaker[, instruction]",
        the others as the VoiceDesign instruction"""
        if self.configuration.tts_model_type == "custom_voice":
            cv = self.parse_custom_voice_prompt(voice)
            return self.prepare_generation_inputs(text, language or "auto", cv[1] if cv else None, cv[0] if cv else None)
        return self.prepare_generation_inputs(text, language or "auto", voice)

    # -- streamingStep / resetStreamingState (Qwen3TTSSpeechTokenizer.swift:948-1006) --------------------------------------
    def set_stream_exact(self, exact: bool):
        """False (default): the reference's streaming arithmetic (bias counted twice after chunk boundaries, :556-559);
        True: chunked decode bitwise equal to decode_codes of the whole sequence."""
        check(_lib.lib().mis_qwen3tts_set_stream_exact(self._h, 1 if exact else 0))

    def reset_streaming_state(self, batch: int = 1, max_frames: int = 4096, max_chunk_frames: int = 64):
        check(_lib.lib().mis_qwen3tts_decode_stream_begin(self._h, batch, max_frames, max_chunk_frames))
        self._stream_batch = batch

    def streaming_step(self, codes) -> np.ndarray:
        """codes [B, num_quantizers, Tn] = only the new frames -> [B, Tn * samples_per_frame]; state stays on the device."""
        cd = np.ascontiguousarray(codes, dtype=np.int32)
        B, nq, T = cd.shape
        if B != getattr(self, "_stream_batch", None):
            raise AudioGenerationError(3, "streaming_step: batch differs from reset_streaming_state")
        out = np.zeros((B, T * self.samples_per_frame), np.float32)
        check(_lib.lib().mis_qwen3tts_decode_stream_step(self._h, cd.ctypes.data, T, out.ctypes.data))
        return out

    def end_streaming(self):
        check(_lib.lib().mis_qwen3tts_decode_stream_end(self._h))

    def generate_stream_batch(self, prompts, generation_parameters: Qwen3TTSGenerateParameters | None = None,
                              streaming_interval: float = 2.0, cancel_flag=None):
        """mis_qwen3tts_generate in streaming mode for a batch of prepared prompts: TokenEvent (code 0 of each frame) and
        AudioEvent chunks of streaming_interval * 12.5 frames WHILE the engine generates, InfoEvent per row when the frame loop
        ends, then the frames after the last full chunk."""
        gp = generation_parameters or self.default_generation_parameters
        t, c, pl, P, tr, tl, Tt = self._marshal(prompts)
        B = len(prompts)
        caps = self._row_caps(prompts, gp)
        gpc = gp.to_c()
        gpc.max_frames = int(caps.max())
        chunk = max(1, int(streaming_interval * 12.5))                  # streamingChunkSize (:394-395)
        pcm = C.c_void_p(); stride = C.c_int64(); plens = (C.c_int64 * B)()

        def start(cbf, flag_addr):
            st = _lib.lib().mis_qwen3tts_generate(self._h, t.ctypes.data, c.ctypes.data, pl.ctypes.data, P, tr.ctypes.data, tl.ctypes.data,
                                                  Tt, B, C.byref(gpc), caps.ctypes.data, C.byref(pcm), C.byref(stride), plens,
                                                  None, None, None, chunk, cbf, None, flag_addr)
            if pcm.value:
                _lib.lib().mis_free(pcm)
            return st
        yield from stream_events(start, decode_audio_event, cancel_flag)

    def generate_stream(self, text: str, voice: str | None = None, language: str | None = None,
                        generation_parameters: Qwen3TTSGenerateParameters | None = None, streaming_interval: float = 2.0):
        """generateStream (:84-133): .token per frame and .audio chunks while generating, .info when the loop ends, then the
        remaining samples."""
        p = self._prepare(text, voice, language)
        yield from self.generate_stream_batch([p], generation_parameters, streaming_interval)
