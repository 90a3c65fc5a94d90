"""Host mirror of SopranoModel : SpeechGenerationModel (Sources/MLXAudioTTS/Models/Soprano/Soprano.swift:201-690).

Device work (Qwen3-style token LM with hidden-state tap, Soprano sampler flavour, Vocos/ISTFT decoder) is behind
mis_soprano_* in libmi_speech.so.  Host logic mirrored here: prompt splitting, sentence merging, the
"[STOP][TEXT]...[START]" framing and the whitespace-aware tokenisation (Soprano.swift:365-575).  Text
normalisation (cleanTextForSoprano, TextUtils.swift) is injectable (`clean_text`) and defaults to a
whitespace collapse: it is string processing outside the accelerated path."""
from __future__ import annotations

import ctypes as C
import re
from dataclasses import dataclass

import numpy as np

from . import _lib
from .codecs import _tensor_args
from .generation import (AudioEvent, AudioGenerationError, AudioGenerationInfo, GenerateParameters, InfoEvent, TokenEvent, check,
                         decode_audio_event, stream_events)
from .tts import MAX_BATCH, LlamaTTSConfiguration, LlamaTTSModel


@dataclass
class SopranoConfiguration:
    """SopranoConfiguration (SopranoConfig.swift:103-167, Soprano-1.1-80M defaults)."""
    hidden_size: int = 512
    num_hidden_layers: int = 17
    intermediate_size: int = 2304
    num_attention_heads: int = 4
    num_key_value_heads: int = 1
    head_dim: int = 128
    vocab_size: int = 8192
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    tie_word_embeddings: bool = False
    sample_rate: int = 32000
    decoder_num_layers: int = 8
    decoder_dim: int = 768
    decoder_intermediate_dim: int = 2304
    hop_length: int = 512
    n_fft: int = 2048
    upscale: int = 4
    input_kernel: int = 1
    dw_kernel: int = 3
    token_size: int = 2048
    stop_token_id: int = 3
    space_token_id: int = 8004           # Soprano.swift:452

    @classmethod
    def from_dict(cls, d: dict) -> "SopranoConfiguration":
        known = {f for f in cls.__dataclass_fields__}
        return cls(**{k: v for k, v in d.items() if k in known})

    def lm_configuration(self) -> LlamaTTSConfiguration:
        return LlamaTTSConfiguration(hidden_size=self.hidden_size, num_hidden_layers=self.num_hidden_layers,
                                     intermediate_size=self.intermediate_size,
                                     num_attention_heads=self.num_attention_heads,
                                     num_key_value_heads=self.num_key_value_heads, head_dim=self.head_dim,
                                     vocab_size=self.vocab_size, rms_norm_eps=self.rms_norm_eps,
                                     rope_theta=self.rope_theta, tie_word_embeddings=self.tie_word_embeddings,
                                     sample_rate=self.sample_rate, qk_norm=True, rope_plain=True)

    def to_c(self) -> "_lib.SopranoConfigC":
        return _lib.SopranoConfigC(self.lm_configuration().to_c(), self.decoder_num_layers, self.decoder_dim,
                                   self.decoder_intermediate_dim, self.hop_length, self.n_fft, self.upscale,
                                   self.input_kernel, self.dw_kernel, self.token_size, self.stop_token_id)


_SENT_SPLIT = re.compile(r"(?<=[.!?])\s+")
_SPECIAL = re.compile(r"\[(?:STOP|TEXT|START)\]")
_PRETOK = re.compile(r"\s+|\w+|[^\w\s]+")


def split_into_sentences(text: str) -> list[str]:
    """Soprano.swift:414-449"""
    return [s for s in _SENT_SPLIT.split(text) if s] or [text]


def preprocess_text(texts, min_length: int = 30, clean_text=None):
    """preprocessText (Soprano.swift:365-411): [(prompt, text_idx, sentence_idx)]."""
    clean = clean_text or (lambda t: re.sub(r"\s+", " ", t).strip())
    out = []
    for ti, text in enumerate(texts):
        items = [s for s in split_into_sentences(clean(text.strip(" \t")))]
        if min_length > 0 and len(items) > 1:
            merged, i = [], 0
            while i < len(items):
                cur = items[i]
                if len(cur) < min_length:
                    if merged:
                        merged[-1] = (merged[-1] + " " + cur).strip(" \t")
                    elif i + 1 < len(items):
                        items[i + 1] = (cur + " " + items[i + 1]).strip(" \t")
                    else:
                        merged.append(cur)
                else:
                    merged.append(cur)
                i += 1
            items = merged
        out += [(f"[STOP][TEXT]{s}[START]", ti, si) for si, s in enumerate(items)]
    return out


def split_prompt(text: str, split_pattern: str = "\n") -> list[str]:
    """Chunking at the top of generate() (Soprano.swift:594-624): split on the pattern, then chunks over 500
    characters at . ? ! : ; once 100 characters are collected (hard cut at 500)."""
    text = text.replace("\\n", "\n").replace("\\t", "\t")
    chunks = [c.strip(" \t") for c in text.split(split_pattern)]
    out = []
    for chunk in (c for c in chunks if c):
        if len(chunk) <= 500:
            out.append(chunk)
            continue
        cur = ""
        for ch in chunk:
            cur += ch
            if (ch in ".?!:;" and len(cur) >= 100) or len(cur) >= 500:
                out.append(cur.strip(" \t"))
                cur = ""
        if cur:
            out.append(cur.strip(" \t"))
    return [c for c in out if c]


class SopranoModel:
    def __init__(self, config: SopranoConfiguration, device: int = 0):
        self.configuration = config
        self.device = device
        self.tokenizer = None
        self.clean_text = None
        self._h = C.c_void_p()
        cc = config.to_c()
        check(_lib.lib().mis_soprano_create(C.byref(cc), device, C.byref(self._h)))
        # borrowed LM handle: taps (lm_forward / reset) for tests and synthetic init
        self.lm = LlamaTTSModel(config.lm_configuration(), None, device,
                                _handle=C.c_void_p(_lib.lib().mis_soprano_lm(self._h)))
        self.lm._borrowed = True

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            if getattr(self, "lm", None) is not None:
                self.lm._h = None
            _lib.lib().mis_soprano_destroy(h)

    @classmethod
    def from_weights(cls, config, weights: dict, device: int = 0) -> "SopranoModel":
        m = cls(config, device)
        for k, v in weights.items():
            m.set_tensor(k, v)
        m.finalize()
        return m

    @classmethod
    def synthetic(cls, config, device: int = 0, seed: int = 4321, decoder_seed: int = 99) -> "SopranoModel":
        from .synthetic import soprano_decoder_synthetic_weights
        m = cls(config, device)
        check(_lib.lib().mis_tts_init_synthetic(m.lm._h, seed))
        for k, v in soprano_decoder_synthetic_weights(config, decoder_seed).items():
            m.set_tensor(k, v)
        m.finalize()
        return m

    def set_tensor(self, name: str, arr):
        keep, ptr, dt, shape = _tensor_args(arr)
        sh = (C.c_int64 * len(shape))(*shape)
        check(_lib.lib().mis_soprano_set_tensor(self._h, name.encode(), ptr, dt, sh, len(shape)))

    def finalize(self):
        check(_lib.lib().mis_soprano_finalize(self._h))

    # -- protocol surface ------------------------------------------------------------------------------
    @property
    def sample_rate(self) -> int:
        return self.configuration.sample_rate

    @property
    def default_generation_parameters(self) -> GenerateParameters:
        """Soprano.swift:231-239"""
        return GenerateParameters(max_tokens=1200, temperature=0.7, top_p=0.95, repetition_penalty=1.5,
                                  repetition_context_size=30, sampler_flavor=1)

    def num_samples(self, n_hidden: int) -> int:
        return int(_lib.lib().mis_soprano_num_samples(self._h, n_hidden))

    def decode(self, hidden) -> np.ndarray:
        """SopranoDecoder.callAsFunction: hidden [B, L, hidden_size] (numpy or device torch) -> [B, samples]."""
        try:
            import torch
            if isinstance(hidden, torch.Tensor):
                hidden = hidden.detach().to(torch.float32).cpu().numpy()
        except ImportError:
            pass
        h = np.ascontiguousarray(hidden, dtype=np.float32)
        B, L, Cc = h.shape
        ptr = h.ctypes.data
        if Cc != self.configuration.hidden_size:
            raise AudioGenerationError(3, "hidden width does not match the configuration")
        out = np.zeros((B, self.num_samples(L)), np.float32)
        check(_lib.lib().mis_soprano_decode(self._h, ptr, B, L, out.ctypes.data))
        return out

    def tokenize(self, text: str) -> np.ndarray:
        """Soprano.swift:461-500: special tokens as-is, one space token per whitespace character."""
        if self.tokenizer is None:
            raise AudioGenerationError(1, "Tokenizer not loaded")
        ids, pos = [], 0
        segs = []
        for m in _SPECIAL.finditer(text):
            if m.start() > pos:
                segs.append((text[pos:m.start()], False))
            segs.append((m.group(0), True))
            pos = m.end()
        if pos < len(text):
            segs.append((text[pos:], False))
        for seg, special in segs:
            if special:
                ids += list(self.tokenizer.encode(seg))
                continue
            for chunk in _PRETOK.findall(seg):
                if chunk.isspace():
                    ids += [self.configuration.space_token_id] * len(chunk)
                else:
                    ids += list(self.tokenizer.encode(chunk))
        return np.asarray(ids, np.int32)

    @property
    def lm_path(self) -> int:
        """mis_soprano_lm_path: the program that ran the LM loop of the last call - 0 launch chain (by rule), 1 batch-1 token engine,
        2 launch chain after the engine's workers could not be co-resident."""
        return int(_lib.lib().mis_soprano_lm_path(self._h))

    def generate_batch(self, prompt_rows, generation_parameters: GenerateParameters | None = None,
                       return_tokens: bool = False, replicas=None):
        """One generate per tokenised sentence prompt, batched: list of 1-D float32 arrays.  More rows than the engine's
        per-call maximum run in slices of MAX_BATCH (the reference loops sentence by sentence, Soprano.swift:637-676);
        the RNG is keyed by the global row index, so slicing does not change any row."""
        gp = generation_parameters or self.default_generation_parameters
        per_call = MAX_BATCH * (len(replicas) if replicas else 1)            # the engine's maximum is per replica
        if len(prompt_rows) > per_call:
            from dataclasses import replace
            outs, toks_all = [], []
            for i in range(0, len(prompt_rows), per_call):
                r = self.generate_batch(prompt_rows[i:i + per_call], replace(gp, row_offset=gp.row_offset + i), return_tokens, replicas)
                if return_tokens:
                    outs += r[0]; toks_all += r[1]
                else:
                    outs += r
            return (outs, toks_all) if return_tokens else outs
        flat, lens = LlamaTTSModel._flatten(prompt_rows)
        B = len(lens)
        gpc = gp.to_c()
        gpc.sampler_flavor = 1
        pcm = C.c_void_p(); stride = C.c_int64(); plens = (C.c_int64 * B)()
        toks = C.c_void_p(); tstride = C.c_int64(); ntok = (C.c_int32 * B)()
        if replicas:          # SopranoModel objects with the same weights, one per GPU: rows sharded inside the library
            hs = (C.c_void_p * len(replicas))(*[r._h for r in replicas])
            check(_lib.lib().mis_soprano_group_generate(hs, len(replicas), flat.ctypes.data, lens.ctypes.data, B, C.byref(gpc), C.byref(pcm),
                                                        C.byref(stride), plens, C.byref(toks) if return_tokens else None,
                                                        C.byref(tstride), ntok))
        else:
            check(_lib.lib().mis_soprano_generate(self._h, flat.ctypes.data, lens.ctypes.data, B, C.byref(gpc), C.byref(pcm),
                                                  C.byref(stride), plens, C.byref(toks) if return_tokens else None,
                                                  C.byref(tstride), ntok))
        try:
            arr = np.ctypeslib.as_array(C.cast(pcm, C.POINTER(C.c_float)), shape=(B, max(stride.value, 1)))
            out = [arr[b, : plens[b]].copy() for b in range(B)]
            if return_tokens:
                t = np.ctypeslib.as_array(C.cast(toks, C.POINTER(C.c_int32)), shape=(B, max(tstride.value, 1)))
                tok = [t[b, : ntok[b]].copy() for b in range(B)]
        finally:
            _lib.lib().mis_free(pcm)
            if return_tokens and toks:
                _lib.lib().mis_free(toks)
        return (out, tok) if return_tokens else out

    def generate(self, text: str, voice=None, split_pattern: str = "\n",
                 generation_parameters: GenerateParameters | None = None) -> np.ndarray:
        """generate(text:voice:splitPattern:parameters:) (Soprano.swift:577-690).  The sentence prompts of the text run
        batched, up to MAX_BATCH per engine call (the reference loops over them); parts are concatenated in order."""
        if self.tokenizer is None:
            raise AudioGenerationError(1, "Tokenizer not loaded")
        gp = generation_parameters or self.default_generation_parameters
        if generation_parameters is None:
            gp.max_tokens = 1200
        rows = []
        for chunk in split_prompt(text, split_pattern):
            rows += [self.tokenize(p) for p, _, _ in preprocess_text([chunk], clean_text=self.clean_text)]
        if not rows:
            raise AudioGenerationError(6, "No audio generated")
        parts = self.generate_batch(rows, gp)
        return np.concatenate(parts) if len(parts) > 1 else parts[0]

    def generate_stream_batch(self, prompt_rows, generation_parameters: GenerateParameters | None = None, cancel_flag=None):
        """mis_soprano_generate_stream on tokenised sentence prompts: TokenEvent while the engine generates, then per row
        InfoEvent and one AudioEvent."""
        gp = generation_parameters or GenerateParameters(max_tokens=512, temperature=0.3, top_p=0.95, repetition_penalty=1.5,
                                                         repetition_context_size=30)          # generateStream defaults (:696-702)
        flat, lens = LlamaTTSModel._flatten(prompt_rows)
        B = len(lens)
        if B > MAX_BATCH:
            raise AudioGenerationError(3, f"at most {MAX_BATCH} sentence prompts per streamed call")
        gpc = gp.to_c()
        gpc.sampler_flavor = 1

        def start(cbf, flag_addr):
            return _lib.lib().mis_soprano_generate_stream(self._h, flat.ctypes.data, lens.ctypes.data, B, C.byref(gpc), cbf, None,
                                                          flag_addr)
        yield from stream_events(start, decode_audio_event, cancel_flag)

    def generate_stream(self, text: str, voice=None, generation_parameters: GenerateParameters | None = None):
        """generateStream(text:voice:parameters:) (Soprano.swift:693-800): .token per sampled id (all sentences of the text, in
        sentence order), then ONE .info and ONE .audio = the sentences' audio concatenated (:760-787).  The sentences run
        sequentially like the reference's loop so that the token order is the reference's."""
        if self.tokenizer is None:
            raise AudioGenerationError(1, "Tokenizer not loaded")
        text = text.replace("\\n", "\n").replace("\\t", "\t")
        rows = [self.tokenize(p) for p, _, _ in preprocess_text([text], clean_text=self.clean_text)]
        parts, total, secs = [], 0, 0.0
        for row in rows:
            for ev in self.generate_stream_batch([row], generation_parameters):
                if isinstance(ev, TokenEvent):
                    yield TokenEvent(0, ev.token)
                elif isinstance(ev, InfoEvent):
                    total += ev.info.generation_token_count
                    secs += ev.info.generate_time
                elif isinstance(ev, AudioEvent):
                    parts.append(ev.audio)
        if not parts:
            raise AudioGenerationError(2, "No audio generated")
        yield InfoEvent(0, AudioGenerationInfo(0, total, 0.0, secs, total / secs if secs > 0 else 0.0, 0.0))
        yield AudioEvent(0, np.concatenate(parts) if len(parts) > 1 else parts[0])
