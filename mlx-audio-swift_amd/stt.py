"""Whisper STT, host mirror of `WhisperModel: STTGenerationModel`
(Sources/MLXAudioSTT/Models/Whisper/WhisperModel.swift:6-309; protocol Sources/MLXAudioSTT/Generation.swift:52-64;
STTOutput Sources/MLXAudioSTT/Models/GLMASR/STTOutput.swift:80-125).  Tokenisation, prompt construction and text
decoding stay on the host, as in the reference (WhisperTokenizer.swift); everything else runs in libmi_speech.so."""
from __future__ import annotations

import ctypes as C
import json
import os
import time
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .codecs import _tensor_args
from .generation import AudioGenerationError, InfoEvent, TokenEvent, check, decode_audio_event, stream_events

SAMPLE_RATE, CHUNK_SAMPLES = 16000, 480000          # WhisperAudioConfig, WhisperConfig.swift:188-193


@dataclass
class WhisperConfig:
    """WhisperConfig.swift:3-22 (both the HF and the OpenAI / mlx-whisper key spellings, :26-55,93-129)."""
    vocab_size: int = 51865
    num_mel_bins: int = 80
    d_model: int = 384
    encoder_layers: int = 4
    encoder_attention_heads: int = 6
    encoder_ffn_dim: int | None = None
    max_source_positions: int = 1500
    decoder_layers: int = 4
    decoder_attention_heads: int = 6
    decoder_ffn_dim: int | None = None
    max_target_positions: int = 448

    @classmethod
    def from_dict(cls, d: dict) -> "WhisperConfig":
        alias = {"n_vocab": "vocab_size", "n_mels": "num_mel_bins", "n_audio_state": "d_model",
                 "n_audio_layer": "encoder_layers", "n_audio_head": "encoder_attention_heads",
                 "n_audio_ctx": "max_source_positions", "n_text_layer": "decoder_layers",
                 "n_text_head": "decoder_attention_heads", "n_text_ctx": "max_target_positions"}
        kw = {}
        for k, v in d.items():
            k = alias.get(k, k)
            if k in cls.__dataclass_fields__ and k not in kw:
                kw[k] = v
        return cls(**kw)

    def to_c(self) -> "_lib.WhisperConfigC":
        return _lib.WhisperConfigC(self.vocab_size, self.num_mel_bins, self.d_model, self.encoder_layers,
                                   self.encoder_attention_heads, self.encoder_ffn_dim or 4 * self.d_model,
                                   self.max_source_positions, self.decoder_layers, self.decoder_attention_heads,
                                   self.decoder_ffn_dim or 4 * self.d_model, self.max_target_positions)


@dataclass
class STTGenerateParameters:
    """Sources/MLXAudioSTT/Generation.swift:3-50 (fields the Whisper loop reads) + WhisperGenerationConfig lists."""
    max_tokens: int = 432
    temperature: float = 0.0
    language: str | None = None
    seed: int = 0
    # tokenizer-owned ids: None = take them from the attached tokenizer (tokenizer.endOfTextId / timestampBeginId,
    # WhisperModel.swift:218,236,238); generate() raises when neither the tokenizer nor the parameters provide them
    eot_id: int | None = None
    timestamp_begin: int | None = None
    suppress_tokens: list | None = None             # None = generationConfig?.suppressTokens ?? [] (:219)
    begin_suppress_tokens: list | None = None       # None = generationConfig?.beginSuppressTokens ?? [eot] (:218)


@dataclass
class STTOutput:
    """STTOutput.swift:80-125"""
    text: str
    segments: list | None
    language: str | None
    prompt_tokens: int
    generation_tokens: int
    total_tokens: int
    prompt_tps: float
    generation_tps: float
    total_time: float
    peak_memory_usage: float
    token_ids: list = field(default_factory=list)   # per chunk (the engine's raw output; text needs a tokenizer)


class WhisperModel:
    """STTGenerationModel conformance: default_generation_parameters, generate(audio, generation_parameters)."""

    def __init__(self, config: WhisperConfig, device: int = 0):
        self.config = config
        self.device = device
        # any object with decode(list[int]) -> str, build_prompt_tokens(language, task) and the attributes end_of_text_id,
        # timestamp_begin_id, is_multilingual, language_to_id (WhisperTokenizer.swift)
        self.tokenizer = None
        self.generation_config = None    # optional dict: suppress_tokens / begin_suppress_tokens (generation_config.json)
        h = C.c_void_p()
        cfg = config.to_c()
        check(_lib.lib().mis_whisper_create(C.byref(cfg), device, C.byref(h)))
        self._h = h

    @classmethod
    def from_weights(cls, config, weights: dict, device: int = 0) -> "WhisperModel":
        m = cls(config, device)
        for name, arr in weights.items():
            m.set_tensor(name, arr)
        m.finalize()
        return m

    @classmethod
    def synthetic(cls, config, device: int = 0, seed: int = 777) -> "WhisperModel":
        m = cls(config, device)
        check(_lib.lib().mis_whisper_init_synthetic(m._h, seed))
        m.finalize()
        return m

    @classmethod
    def from_model_directory(cls, model_dir: str, device: int = 0) -> "WhisperModel":
        """fromDirectory: config.json + *.safetensors (WhisperModel.swift:337-363) in either key layout - HF transformers
        ("model.encoder.layers.N.self_attn.q_proj.*") or OpenAI / mlx-whisper ("encoder.blocks.N.attn.query.*", MLX conv
        layout, no encoder positional embedding); the engine's set_tensor applies WhisperModel.sanitize (:321-478)."""
        from safetensors import safe_open
        with open(os.path.join(model_dir, "config.json")) as f:
            cfg = WhisperConfig.from_dict(json.load(f))
        m = cls(cfg, device)
        for fn in sorted(os.listdir(model_dir)):
            if fn.endswith(".safetensors"):
                with safe_open(os.path.join(model_dir, fn), framework="pt") as sf:
                    for k in sf.keys():
                        m.set_tensor(k, sf.get_tensor(k))
        m.finalize()
        gc = os.path.join(model_dir, "generation_config.json")
        if os.path.exists(gc):
            with open(gc) as f:
                m.generation_config = json.load(f)
        return m

    def set_tensor(self, name: str, arr):
        keep, ptr, dt, shape = _tensor_args(arr)
        sh = (C.c_int64 * len(shape))(*shape)
        check(_lib.lib().mis_whisper_set_tensor(self._h, name.encode(), ptr, dt, sh, len(shape)))

    def finalize(self):
        check(_lib.lib().mis_whisper_finalize(self._h))

    @property
    def default_generation_parameters(self) -> STTGenerateParameters:      # WhisperModel.swift:21-34
        return STTGenerateParameters(max_tokens=self.config.max_target_positions - 16)

    # -- taps for parity tests -----------------------------------------------------------------------
    def encode(self, features: np.ndarray, want_output: bool = True):
        f = np.ascontiguousarray(features, dtype=np.float32)
        B = f.shape[0]
        out = np.zeros((B, 1500, self.config.d_model), np.float32) if want_output else None
        check(_lib.lib().mis_whisper_encode(self._h, f.ctypes.data, B, out.ctypes.data if want_output else None))
        return out

    def decoder_reset(self):
        check(_lib.lib().mis_whisper_decoder_reset(self._h))

    def decoder_forward(self, tokens, active=None, want_logits: bool = True):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        B = t.shape[0]
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        out = np.zeros((B, self.config.vocab_size), np.float32) if want_logits else None
        check(_lib.lib().mis_whisper_decoder_forward(self._h, t.ctypes.data, act.ctypes.data if act is not None else None,
                                                     out.ctypes.data if want_logits else None))
        return out

    # -- generate ------------------------------------------------------------------------------------
    def transcribe_windows(self, windows, prompt_ids, params: STTGenerateParameters, replicas=None):
        """transcribeChunk for a batch of <= 30 s windows: list of 1-D float arrays -> list of token-id lists.
        replicas: WhisperModel objects holding the same weights, one per GPU (self included): the windows are sharded over them
        inside the library (mis_whisper_group_generate) and the token ids gathered in window order."""
        B = len(windows)
        stride = max(1, max(len(w) for w in windows))
        pcm = np.zeros((B, stride), np.float32)
        lens = np.zeros(B, np.int64)
        for i, w in enumerate(windows):
            w = np.asarray(w, np.float32).reshape(-1)
            pcm[i, : len(w)] = w
            lens[i] = len(w)
        params = self._resolve_parameters(params)
        prompt = np.ascontiguousarray(prompt_ids, dtype=np.int32)
        sup = np.ascontiguousarray(params.suppress_tokens or [], dtype=np.int32)
        bs = params.begin_suppress_tokens if params.begin_suppress_tokens is not None else [params.eot_id]
        bsup = np.ascontiguousarray(bs, dtype=np.int32)
        sp = _lib.SttParamsC(int(params.max_tokens), float(params.temperature), int(params.seed), int(params.eot_id),
                             int(params.timestamp_begin), sup.ctypes.data if len(sup) else None, len(sup),
                             bsup.ctypes.data if len(bsup) else None, len(bsup))
        toks = C.c_void_p(); ts = C.c_int64(); nt = (C.c_int32 * B)()
        if replicas:
            hs = (C.c_void_p * len(replicas))(*[r._h for r in replicas])
            check(_lib.lib().mis_whisper_group_generate(hs, len(replicas), pcm.ctypes.data, lens.ctypes.data, B, stride, prompt.ctypes.data,
                                                        len(prompt), C.byref(sp), C.byref(toks), C.byref(ts), nt))
        else:
            check(_lib.lib().mis_stt_whisper_generate(self._h, pcm.ctypes.data, lens.ctypes.data, B, stride, prompt.ctypes.data,
                                                      len(prompt), C.byref(sp), C.byref(toks), C.byref(ts), nt))
        try:
            arr = np.ctypeslib.as_array(C.cast(toks, C.POINTER(C.c_int32)), shape=(B, max(ts.value, 1)))
            return [arr[b, : nt[b]].tolist() for b in range(B)]
        finally:
            _lib.lib().mis_free(toks)

    def _resolve_parameters(self, gp: STTGenerateParameters) -> STTGenerateParameters:
        """transcribeChunk reads the end-of-text id, the first timestamp id and the default suppress lists from the tokenizer /
        generation config (WhisperModel.swift:218-219,236-238), never from the caller: fill what the parameters leave open."""
        from dataclasses import replace
        tk, gc = self.tokenizer, (self.generation_config or {})
        eot = gp.eot_id if gp.eot_id is not None else getattr(tk, "end_of_text_id", None)
        tsb = gp.timestamp_begin if gp.timestamp_begin is not None else getattr(tk, "timestamp_begin_id", None)
        if eot is None or tsb is None:
            raise AudioGenerationError(3, "end-of-text / timestamp-begin ids unknown: attach a tokenizer (end_of_text_id, "
                                          "timestamp_begin_id) or set eot_id and timestamp_begin in STTGenerateParameters")
        # `generationConfig?.suppressTokens ?? []`, `generationConfig?.beginSuppressTokens ?? [eot]` (:218-219): only a MISSING key takes the
        # default - an explicit empty list in generation_config.json means "suppress nothing"
        gsup, gbsup = gc.get("suppress_tokens"), gc.get("begin_suppress_tokens")
        sup = gp.suppress_tokens if gp.suppress_tokens is not None else list(gsup if gsup is not None else [])
        bsup = gp.begin_suppress_tokens if gp.begin_suppress_tokens is not None else list(gbsup if gbsup is not None else [eot])
        return replace(gp, eot_id=int(eot), timestamp_begin=int(tsb), suppress_tokens=sup, begin_suppress_tokens=bsup)

    def _prompt_language(self, prompt_ids, fallback):
        """The language token sits at prompt index 1 for multilingual models (WhisperModel.swift:271-280: nil otherwise); `fallback` is the
        caller's `detectedLanguage ?? generationParameters.language` (:81, :144)."""
        tk = self.tokenizer
        if tk is not None and getattr(tk, "is_multilingual", False) and len(prompt_ids) > 1:
            for code, tid in getattr(tk, "language_to_id", {}).items():
                if tid == int(prompt_ids[1]):
                    return code
        return fallback

    def generate(self, audio, generation_parameters: STTGenerateParameters | None = None, prompt_ids=None,
                 max_batch: int = 64) -> STTOutput:
        """generate(audio:generationParameters:) (WhisperModel.swift:36-90): mono mix, hard 30 s chunking (:165-182),
        every chunk transcribed (here: batched on the device), texts joined with spaces."""
        gp = self._resolve_parameters(generation_parameters or self.default_generation_parameters)
        t0 = time.time()
        a = np.asarray(audio, np.float32)
        mono = a.mean(axis=-1) if a.ndim > 1 else a
        chunks = [mono] if len(mono) <= CHUNK_SAMPLES else [mono[i:i + CHUNK_SAMPLES] for i in range(0, len(mono), CHUNK_SAMPLES)]
        if prompt_ids is None:
            if self.tokenizer is None:
                raise AudioGenerationError(1, "WhisperTokenizer not loaded")
            prompt_ids = self.tokenizer.build_prompt_tokens(language=gp.language, task="transcribe")
        ids = []
        for i in range(0, len(chunks), max_batch):
            ids += self.transcribe_windows(chunks[i:i + max_batch], prompt_ids, gp)
        texts, segments = [], []
        for ci, tok in enumerate(ids):
            text = self.tokenizer.decode(tok).strip() if self.tokenizer is not None else ""
            if text:
                texts.append(text)
                start = ci * CHUNK_SAMPLES / SAMPLE_RATE
                segments.append({"text": text, "start": start, "end": start + len(chunks[ci]) / SAMPLE_RATE})
        el = time.time() - t0
        n_prompt, n_gen = len(prompt_ids) * len(chunks), sum(len(t) for t in ids)
        return STTOutput(" ".join(texts), segments or None, self._prompt_language(prompt_ids, gp.language), n_prompt, n_gen, n_prompt + n_gen,
                         n_prompt / el if el > 0 else 0.0, n_gen / el if el > 0 else 0.0, el, 0.0, ids)

    def transcribe_windows_stream(self, windows, prompt_ids, params: STTGenerateParameters, cancel_flag=None):
        """mis_stt_whisper_generate_stream: yields TokenEvent(row, id) while the greedy loop runs, then InfoEvent per row."""
        B = len(windows)
        stride = max(1, max(len(w) for w in windows))
        pcm = np.zeros((B, stride), np.float32)
        lens = np.zeros(B, np.int64)
        for i, w in enumerate(windows):
            w = np.asarray(w, np.float32).reshape(-1)
            pcm[i, : len(w)] = w
            lens[i] = len(w)
        params = self._resolve_parameters(params)
        prompt = np.ascontiguousarray(prompt_ids, dtype=np.int32)
        sup = np.ascontiguousarray(params.suppress_tokens or [], dtype=np.int32)
        bsup = np.ascontiguousarray(params.begin_suppress_tokens, dtype=np.int32)
        sp = _lib.SttParamsC(int(params.max_tokens), float(params.temperature), int(params.seed), int(params.eot_id),
                             int(params.timestamp_begin), sup.ctypes.data if len(sup) else None, len(sup),
                             bsup.ctypes.data if len(bsup) else None, len(bsup))

        def start(cbf, flag_addr):
            return _lib.lib().mis_stt_whisper_generate_stream(self._h, pcm.ctypes.data, lens.ctypes.data, B, stride, prompt.ctypes.data,
                                                              len(prompt), C.byref(sp), cbf, None, flag_addr, None, None, None)
        yield from stream_events(start, decode_audio_event, cancel_flag)

    def generate_stream(self, audio, generation_parameters: STTGenerateParameters | None = None, prompt_ids=None):
        """generateStream(audio:generationParameters:) (WhisperModel.swift:92-160): chunk by chunk, yields ("token", text delta)
        per step whose decoded text changed (decode-and-diff, :242-254) and finally ("result", STTOutput)."""
        gp = self._resolve_parameters(generation_parameters or self.default_generation_parameters)
        if self.tokenizer is None:
            raise AudioGenerationError(1, "WhisperTokenizer not loaded")
        t0 = time.time()
        a = np.asarray(audio, np.float32)
        mono = a.mean(axis=-1) if a.ndim > 1 else a
        chunks = [mono] if len(mono) <= CHUNK_SAMPLES else [mono[i:i + CHUNK_SAMPLES] for i in range(0, len(mono), CHUNK_SAMPLES)]
        if prompt_ids is None:
            prompt_ids = self.tokenizer.build_prompt_tokens(language=gp.language, task="transcribe")
        texts, segments, ids = [], [], []
        for ci, chunk in enumerate(chunks):                       # the reference streams chunk after chunk (:107-121)
            generated, previous = [], ""
            for ev in self.transcribe_windows_stream([chunk], prompt_ids, gp):
                if isinstance(ev, TokenEvent):
                    generated.append(ev.token)
                    so_far = self.tokenizer.decode(generated)
                    if so_far != previous:
                        delta = so_far[len(previous):] if so_far.startswith(previous) else so_far
                        previous = so_far
                        if delta:
                            yield ("token", delta)
            ids.append(generated)
            text = self.tokenizer.decode(generated).strip()
            if text:
                texts.append(text)
                start = ci * CHUNK_SAMPLES / SAMPLE_RATE
                segments.append({"text": text, "start": start, "end": start + len(chunk) / SAMPLE_RATE})
        el = time.time() - t0
        n_prompt, n_gen = len(prompt_ids) * len(chunks), sum(len(t) for t in ids)
        yield ("result", STTOutput(" ".join(texts), segments or None, self._prompt_language(prompt_ids, gp.language), n_prompt, n_gen,
                                   n_prompt + n_gen, n_prompt / el if el > 0 else 0.0, n_gen / el if el > 0 else 0.0, el, 0.0, ids))

    def close(self):
        if self._h is not None:
            _lib.lib().mis_whisper_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
