"""Orpheus / Llama TTS, host mirror of `class LlamaTTSModel: SpeechGenerationModel`
(Sources/MLXAudioTTS/Models/Llama/LlamaTTS.swift:354-977).  Tokenisation stays on the host, as in
the reference (swift-transformers, LlamaTTS.swift:487); everything after token ids runs in
libmi_speech.so."""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .codecs import SNAC, _tensor_args
from .generation import (AudioEvent, AudioGenerationError, AudioGenerationInfo, GenerateParameters, InfoEvent,
                         TokenEvent, check, decode_audio_event, stream_events)


MAX_BATCH = 64          # rows per engine call ("batch per GPU must be 1..64", csrc/lm_engine.hip)


class OrpheusTokens:
    """LlamaTTS.swift:20-30"""
    start_of_human = 128259
    end_of_human = 128260
    end_of_text = 128009
    start_of_speech = 128257
    end_of_speech = 128258
    pad_token = 128263
    audio_start = 128261
    audio_end = 128262
    audio_token_offset = 128266


class VyvoTokens:
    """VyvoTTS (Qwen3 + SNAC) token ids, Sources/MLXAudioTTS/Models/Qwen3/Qwen3.swift:19-29"""
    tokenizer_length = 151669
    start_of_text = 151643
    end_of_text = 151645
    start_of_speech = 151670
    end_of_speech = 151671
    start_of_human = 151672
    end_of_human = 151673
    start_of_ai = 151674
    end_of_ai = 151675
    pad_token = 151676
    audio_token_offset = 151679
    codec_chunk_groups = 50          # decodeAudioFromCodes(chunkSize:), Qwen3.swift:47


# -- host-side integer framing of the Orpheus prompt (no device involved; tests/test_tts_host_cpu.py) ---------------------------------
def interleave_snac_codes(l1, l2, l3) -> np.ndarray:
    """The interleave of llamaEncodeAudioToCodes (LlamaTTS.swift:80-97): three SNAC code levels of one utterance (g, 2g, 4g codes)
    -> g frames of 7 ids [l1 | l2+4096 | l3+2*4096 | l3+3*4096 | l2+4*4096 | l3+5*4096 | l3+6*4096]; the exact inverse of the decode-side
    de-interleave (:41-64)."""
    l1, l2, l3 = (np.asarray(a).astype(np.int64).reshape(-1) for a in (l1, l2, l3))
    g = len(l1)
    if len(l2) != 2 * g or len(l3) != 4 * g:
        raise AudioGenerationError(3, f"SNAC code levels of lengths {len(l1)}, {len(l2)}, {len(l3)} are not a 1:2:4 hierarchy")
    out = np.empty((g, 7), np.int64)
    out[:, 0] = l1
    out[:, 1] = l2[0::2] + 4096
    out[:, 2] = l3[0::4] + 2 * 4096
    out[:, 3] = l3[1::4] + 3 * 4096
    out[:, 4] = l2[1::2] + 4 * 4096
    out[:, 5] = l3[2::4] + 5 * 4096
    out[:, 6] = l3[3::4] + 6 * 4096
    return out.reshape(-1).astype(np.int32)


def orpheus_prompt_rows(tokenizer, prompts, voice=None, ref_codes=None, ref_text=None, tokens=OrpheusTokens):
    """The id rows of prepareInputIds (LlamaTTS.swift:446-553), one int32 array per prompt:
        [SOH ref-transcript EOT EOH AUDIO_START START_OF_SPEECH ref-codes+offset END_OF_SPEECH AUDIO_END]?  SOH text EOT EOH
    The bracketed part appears only with BOTH `ref_codes` (the interleaved SNAC codes of the reference recording, before the
    audio-token offset, :466-467) and `ref_text` (:458); `voice` prefixes every prompt with "voice: " (:472-476).  The reference's
    explicit pad tokens are not materialised here: see `padded_prompt_batch`."""
    T = tokens
    ref = []
    if ref_codes is not None and ref_text is not None:                    # :457-469, :505-528
        audio_ids = np.asarray(ref_codes).astype(np.int64).reshape(-1) + T.audio_token_offset
        ref = ([T.start_of_human] + [int(t) for t in tokenizer.encode(ref_text)] + [T.end_of_text, T.end_of_human] +
               [T.audio_start, T.start_of_speech] + [int(t) for t in audio_ids] + [T.end_of_speech, T.audio_end])
    rows = []
    for p in prompts:
        if voice is not None:
            p = f"{voice}: {p}"
        ids = [int(t) for t in tokenizer.encode(p)]
        rows.append(np.asarray(ref + [T.start_of_human] + ids + [T.end_of_text, T.end_of_human], np.int32))
    return rows


def padded_prompt_batch(rows, pad_token: int = OrpheusTokens.pad_token):
    """What prepareInputIds returns (LlamaTTS.swift:495-552): the rows LEFT-padded with the pad token to the longest row as one
    [batch, maxLen] int32 matrix and the mask `ids != pad`.  (The reference pads by the PROMPT lengths, and every row carries the same
    reference prefix, so that is the same as padding by the row lengths.)  The engine takes the ragged rows and masks the padding
    itself; this form is for callers that want the reference's return value."""
    n = max((len(r) for r in rows), default=0)
    ids = np.full((len(rows), n), pad_token, np.int32)
    for i, r in enumerate(rows):
        if len(r):
            ids[i, n - len(r):] = r
    return ids, ids != pad_token


@dataclass
class LlamaTTSConfiguration:
    """LlamaTTSConfig.swift:15-61 (CodingKeys = HF config.json names)."""
    hidden_size: int = 3072
    num_hidden_layers: int = 28
    intermediate_size: int = 8192
    num_attention_heads: int = 24
    num_key_value_heads: int | None = 8
    head_dim: int | None = 128
    rms_norm_eps: float = 1e-5
    vocab_size: int = 156940
    rope_theta: float = 10000.0
    rope_traditional: bool = False
    rope_scaling: dict | None = None
    tie_word_embeddings: bool = True
    attention_bias: bool = False
    mlp_bias: bool = False
    sample_rate: int = 24000
    max_position_embeddings: int | None = None
    qk_norm: bool = False          # Qwen3-style LMs (Soprano, VyvoTTS): per-head q/k RMSNorm ...
    rope_plain: bool = False       # ... and RoPE(base) without the llama3 rescale
    rope_ops_in_dtype: bool = False  # Qwen3-TTS: rotation as bf16 array ops (Qwen3TTSTalker.swift:15-24)
    # speech token ids of the generate loop; 0 = Orpheus (OrpheusTokens).  VyvoTTS: VyvoTokens
    start_of_speech_id: int = 0
    end_of_speech_id: int = 0
    audio_token_offset: int = 0
    start_of_ai_id: int = 0
    codec_chunk_groups: int = 0    # VyvoTTS: 50 (independent SNAC chunks, Qwen3.swift:47-83); 0 = one decode per utterance

    @classmethod
    def from_dict(cls, d: dict) -> "LlamaTTSConfiguration":
        cfg = cls(**{k: d[k] for k in cls.__dataclass_fields__ if k in d})
        rs = cfg.rope_scaling
        if rs is not None:                                   # validation of LlamaTTSConfig.swift:139-165
            if "factor" not in rs:
                raise AudioGenerationError(3, "rope_scaling must contain 'factor'")
            if "type" not in rs and "rope_type" not in rs:
                raise AudioGenerationError(3, "rope_scaling must contain either 'type' or 'rope_type'")
        return cfg

    def to_c(self) -> "_lib.LmConfigC":
        if self.rope_traditional or self.attention_bias or self.mlp_bias:
            raise AudioGenerationError(3, "rope_traditional / attention_bias / mlp_bias variants are not supported")
        rs = self.rope_scaling or {}
        return _lib.LmConfigC(self.hidden_size, self.num_hidden_layers, self.intermediate_size,
                              self.num_attention_heads, self.num_key_value_heads or self.num_attention_heads,
                              self.head_dim or 0, self.vocab_size, self.rms_norm_eps, self.rope_theta,
                              float(rs.get("factor", 32.0)), float(rs.get("low_freq_factor", 1.0)),
                              float(rs.get("high_freq_factor", 4.0)),
                              float(rs.get("original_max_position_embeddings", 8192.0)),
                              1 if self.tie_word_embeddings else 0, self.sample_rate,
                              1 if self.qk_norm else 0, 1 if self.rope_plain else 0, 1 if self.rope_ops_in_dtype else 0,
                              self.start_of_speech_id, self.end_of_speech_id, self.audio_token_offset, self.start_of_ai_id,
                              self.codec_chunk_groups)


class LlamaTTSModel:
    """SpeechGenerationModel conformance: sample_rate, default_generation_parameters, generate,
    generate_stream (Generation.swift:8-39).  Batch methods (`generate_batch`) are the new capability
    (the reference is batch-1, LlamaTTS.swift:683-688); row r of a batch equals the B=1 result."""

    def __init__(self, config: LlamaTTSConfiguration, codec: SNAC | None = None, device: int = 0, _handle=None):
        self.configuration = config
        self.device = device
        self._snac_model = codec
        self.tokenizer = None
        self._h = _handle
        if self._h is None:
            h = C.c_void_p()
            cfg = config.to_c()
            check(_lib.lib().mis_tts_create(C.byref(cfg), codec._h if codec else None, device, C.byref(h)))
            self._h = h

    # -- loading (LlamaTTS.swift:915-993) ---------------------------------------------------------
    @classmethod
    def from_model_directory(cls, model_dir: str, codec: SNAC | None = None, device: int = 0) -> "LlamaTTSModel":
        with open(os.path.join(model_dir, "config.json")) as f:
            cfg = LlamaTTSConfiguration.from_dict(json.load(f))
        h = C.c_void_p()
        check(_lib.lib().mis_tts_load(model_dir.encode(), codec._h if codec else None, device, C.byref(h)))
        return cls(cfg, codec, device, _handle=h)

    @classmethod
    def from_pretrained(cls, model_repo: str, codec: SNAC | None = None, device: int = 0) -> "LlamaTTSModel":
        if os.path.isdir(model_repo):
            return cls.from_model_directory(model_repo, codec, device)
        raise AudioGenerationError(1, f"model repo {model_repo!r} is not a local directory (no network access)")

    @classmethod
    def from_weights(cls, config, weights: dict, codec: SNAC | None = None, device: int = 0) -> "LlamaTTSModel":
        m = cls(config, codec, device)
        for name, arr in weights.items():
            m.set_tensor(name, arr)
        m.finalize()
        return m

    @classmethod
    def synthetic(cls, config, codec: SNAC | None = None, device: int = 0, seed: int = 4321, quant_bits: int | None = None) -> "LlamaTTSModel":
        """Random weights generated on the device (mis-synth-v1); there are no checkpoints offline.  quant_bits 8 / 4: every
        Linear as a synthetic MLX affine-quantised matrix (group 64), streamed in that form."""
        m = cls(config, codec, device)
        if quant_bits:
            check(_lib.lib().mis_tts_init_synthetic_quantized(m._h, seed, int(quant_bits)))
        else:
            check(_lib.lib().mis_tts_init_synthetic(m._h, seed))
        m.finalize()
        return m

    @property
    def native_quant_bits(self) -> dict:
        """Per role, the bit width of the quantised form it is streamed in (0 = dense bf16)."""
        return {r: int(_lib.lib().mis_tts_native_quant_bits(self._h, i)) for i, r in enumerate(("qkv", "o", "gate_up", "down", "lm_head"))}

    def set_tensor(self, name: str, arr):
        keep, ptr, dt, shape = _tensor_args(arr)
        sh = (C.c_int64 * len(shape))(*shape)
        check(_lib.lib().mis_tts_set_tensor(self._h, name.encode(), ptr, dt, sh, len(shape)))

    def set_quantized_tensor(self, name: str, wq, scales, biases, group_size: int = 64, bits: int = 4):
        """A matrix in MLX's affine-quantised form: wq uint32 [N, K*bits/32], scales / biases [N, K/group_size]."""
        wq = np.ascontiguousarray(wq, dtype=np.uint32)
        ks, ps, ds, ss = _tensor_args(scales)
        kb, pb, db, sb = _tensor_args(biases)
        if ds != db or tuple(ss) != tuple(sb):
            raise AudioGenerationError(3, "scales and biases must share dtype and shape")
        N, K = int(ss[0]), int(ss[1]) * group_size
        check(_lib.lib().mis_tts_set_tensor_quantized(self._h, name.encode(), wq.ctypes.data, ps, pb, ds, N, K, group_size, bits))

    def finalize(self):
        check(_lib.lib().mis_tts_finalize(self._h))

    # -- protocol surface --------------------------------------------------------------------------
    @property
    def sample_rate(self) -> int:
        return self.configuration.sample_rate

    @property
    def default_generation_parameters(self) -> GenerateParameters:
        return GenerateParameters()

    def encode_audio_to_codes(self, audio) -> np.ndarray:
        """llamaEncodeAudioToCodes (LlamaTTS.swift:72-98): SNAC-encode a reference waveform and interleave the three code
        levels into 7-token frames with the k*4096 slot offsets (inverse of the decode-side de-interleave, :41-64)."""
        if self._snac_model is None:
            raise AudioGenerationError(1, "SNAC model not loaded")
        l1, l2, l3 = [c[0] for c in self._snac_model.encode(np.asarray(audio, np.float32).reshape(1, -1))]
        return interleave_snac_codes(l1, l2, l3)

    def prepare_input_ids(self, prompts, voice=None, ref_audio=None, ref_text=None):
        """prepareInputIds (LlamaTTS.swift:446-553) as the list of per-row id arrays (`orpheus_prompt_rows`); the engine left-pads
        the rows itself, `padded_prompt_batch(rows)` gives the reference's (ids, mask) pair.  Needs `self.tokenizer` (any object
        with .encode)."""
        if self.tokenizer is None:
            raise AudioGenerationError(1, "Tokenizer not loaded")
        ref_codes = None
        if ref_audio is not None and ref_text is not None:                # voice cloning branch (:457-469)
            ref_codes = self.encode_audio_to_codes(ref_audio)
        return orpheus_prompt_rows(self.tokenizer, prompts, voice, ref_codes, ref_text)

    def generate(self, text: str, voice=None, ref_audio=None, ref_text=None, language=None,
                 generation_parameters: GenerateParameters | None = None, snac_noise=None) -> np.ndarray:
        """generate(text:voice:...) -> 1-D float32 PCM (LlamaTTS.swift:658-765)."""
        text = text.replace("\\n", "\n").replace("\\t", "\t")            # :680-681
        rows = self.prepare_input_ids([text], voice, ref_audio, ref_text)
        return self.generate_batch(rows, generation_parameters, snac_noise)[0]

    def generate_batch(self, prompt_rows, generation_parameters: GenerateParameters | None = None, snac_noise=None,
                       return_tokens: bool = False):
        """Batched generate on already-tokenised prompts: list of 1-D float32 arrays (one per row).  More than MAX_BATCH
        rows run in slices (RNG keyed by the global row index: slicing changes no row)."""
        gp = generation_parameters or self.default_generation_parameters
        if len(prompt_rows) > MAX_BATCH:
            if snac_noise is not None:
                raise AudioGenerationError(3, f"explicit SNAC noise is limited to {MAX_BATCH} rows per call")
            from dataclasses import replace
            outs, toks_all = [], []
            for i in range(0, len(prompt_rows), MAX_BATCH):
                r = self.generate_batch(prompt_rows[i:i + MAX_BATCH], replace(gp, row_offset=gp.row_offset + i), None, return_tokens)
                if return_tokens:
                    outs += r[0]; toks_all += r[1]
                else:
                    outs += r
            return (outs, toks_all) if return_tokens else outs
        flat, lens = self._flatten(prompt_rows)
        B = len(lens)
        gpc = gp.to_c()
        pcm = C.c_void_p(); stride = C.c_int64(); plens = (C.c_int64 * B)()
        toks = C.c_void_p(); tstride = C.c_int64(); ntok = (C.c_int32 * B)()
        nptr, keep = self._noise_ptrs(snac_noise)
        check(_lib.lib().mis_tts_generate(self._h, flat.ctypes.data, lens.ctypes.data, B, C.byref(gpc), nptr,
                                          C.byref(pcm), C.byref(stride), plens,
                                          C.byref(toks) if return_tokens else None, C.byref(tstride), ntok))
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

    def generate_stream(self, text: str, voice=None, ref_audio=None, ref_text=None, language=None,
                        generation_parameters: GenerateParameters | None = None, snac_noise=None):
        """generateStream(...) (LlamaTTS.swift:777-913): yields TokenEvent per step, then InfoEvent and
        ONE final AudioEvent (Orpheus does not stream audio chunks)."""
        text = text.replace("\\n", "\n").replace("\\t", "\t")
        rows = self.prepare_input_ids([text], voice, ref_audio, ref_text)        # :806-811 passes refAudio / refText too
        yield from self.generate_stream_batch(rows, generation_parameters, snac_noise)

    def generate_stream_batch(self, prompt_rows, generation_parameters=None, snac_noise=None, cancel_flag=None):
        """Events are yielded while the engine is still generating (the C call runs on a worker thread); closing the
        generator cancels the generation at the next poll of the decode loop."""
        gp = generation_parameters or self.default_generation_parameters
        flat, lens = self._flatten(prompt_rows)
        B = len(lens)
        gpc = gp.to_c()
        nptr, keep = self._noise_ptrs(snac_noise)

        def start(cbf, flag_addr):
            return _lib.lib().mis_tts_generate_stream(self._h, flat.ctypes.data, lens.ctypes.data, B, C.byref(gpc), nptr, cbf,
                                                      None, flag_addr)
        yield from stream_events(start, decode_audio_event, cancel_flag)

    # -- LM taps used by the parity tests ------------------------------------------------------------
    def lm_reset(self, batch: int, max_context: int):
        check(_lib.lib().mis_lm_reset(self._h, batch, max_context))

    def lm_forward(self, ids, active=None, want_logits: bool = True, want_hidden: bool = False):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        B = ids.shape[0]
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        out = np.zeros((B, self.configuration.vocab_size), np.float32) if want_logits else None
        hid = np.zeros((B, self.configuration.hidden_size), np.float32) if want_hidden else None
        check(_lib.lib().mis_lm_forward_hidden(self._h, ids.ctypes.data, act.ctypes.data if act is not None else None,
                                               out.ctypes.data if want_logits else None,
                                               hid.ctypes.data if want_hidden else None))
        return (out, hid) if want_hidden else out

    def debug_token_engine(self, prompt, n_new: int, xcds: int = 2, want_logits: bool = False, want_hidden: bool = False,
                           sampling: GenerateParameters | None = None, stop_id: int = -1):
        """csrc/token_engine.hip through include/mi_speech_debug.h: the whole batch-1 request in ONE persistent launch on the compute
        units of `xcds` XCDs.  sampling=None: arg-max after every position (rows indexed by position).  sampling given: the Soprano
        loop's semantics - rows from the last prompt position on, until stop_id or n_new ids.  Returns a dict: next_tokens, ms,
        positions, chosen, logits / hidden when asked for."""
        prompt = np.ascontiguousarray(prompt, dtype=np.int32)
        n = len(prompt) + int(n_new)
        rows = n if sampling is None else int(n_new) + 1
        nxt = np.zeros(n, np.int32)
        lg = np.zeros((rows if sampling is None else max(int(n_new), 1), self.configuration.vocab_size), np.float32) if want_logits else None
        hid = np.zeros((rows, self.configuration.hidden_size), np.float32) if want_hidden else None
        ms = C.c_double(0.0)
        counts = (C.c_int32 * 2)()
        gpc = sampling.to_c() if sampling is not None else None
        check(_lib.lib().mis_debug_token_engine(self._h, prompt.ctypes.data, len(prompt), int(n_new), int(xcds),
                                                C.byref(gpc) if gpc is not None else None, int(stop_id), nxt.ctypes.data,
                                                lg.ctypes.data if want_logits else None, hid.ctypes.data if want_hidden else None,
                                                counts, C.byref(ms)))
        return {"next_tokens": nxt, "ms": ms.value, "logits": lg, "hidden": hid, "positions": int(counts[0]), "chosen": int(counts[1])}

    def last_timing(self) -> dict:
        t = _lib.TimingC()
        check(_lib.lib().mis_tts_last_timing(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def lm_prefill(self, rows, max_context: int = 0, want_logits: bool = True, want_hidden: bool = False):
        """The prefill call `model(inputIds, cache:)` (LlamaTTS.swift:711) for ragged prompts: next-token logits [B, V] (and / or
        model.norm(h) of every row's last prompt token [B, d]); the caches then hold the prompts and lm_forward continues behind them."""
        flat, lens = self._flatten(rows)
        B = len(lens)
        out = np.zeros((B, self.configuration.vocab_size), np.float32) if want_logits else None
        hid = np.zeros((B, self.configuration.hidden_size), np.float32) if want_hidden else None
        check(_lib.lib().mis_lm_prefill(self._h, flat.ctypes.data, lens.ctypes.data, B, int(max_context),
                                        out.ctypes.data if want_logits else None, hid.ctypes.data if want_hidden else None))
        return (out, hid) if want_hidden else out

    def time_gemm(self, which: int, batch: int, iters: int = 20):
        ms, by = C.c_double(), C.c_double()
        check(_lib.lib().mis_tts_time_gemm(self._h, which, batch, iters, C.byref(ms), C.byref(by)))
        return ms.value, by.value

    # -- helpers ---------------------------------------------------------------------------------------
    @staticmethod
    def _flatten(rows):
        rows = [np.ascontiguousarray(r, dtype=np.int32).ravel() for r in rows]
        if not rows:
            raise AudioGenerationError(3, "empty batch")
        lens = np.asarray([len(r) for r in rows], np.int32)
        return np.ascontiguousarray(np.concatenate(rows)), lens

    @staticmethod
    def _noise_ptrs(noise):
        if noise is None:
            return None, None
        keep = [np.ascontiguousarray(n, dtype=np.float32) for n in noise]
        return (C.c_void_p * len(keep))(*[n.ctypes.data for n in keep]), keep

    def close(self):
        if self._h is not None:
            if not getattr(self, "_borrowed", False):      # SopranoModel owns its LM handle
                _lib.lib().mis_tts_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sample_logits(logits, window, window_len, params: GenerateParameters, step: int, lo: int = 0, hi: int = 0,
                  device: int = 0):
    """Stand-alone processor+sampler (mis_sample_logits) for parity tests."""
    l = np.ascontiguousarray(logits, dtype=np.float32)
    B, V = l.shape
    w = np.ascontiguousarray(window, dtype=np.int32)
    ctx = w.shape[1] if w.size else 0
    wl = np.ascontiguousarray(window_len, dtype=np.int32)
    out = np.zeros(B, np.int32)
    gpc = params.to_c()
    check(_lib.lib().mis_sample_logits(device, l.ctypes.data, B, V, w.ctypes.data if ctx else None,
                                       wl.ctypes.data if ctx else None, ctx, C.byref(gpc), step, lo, hi, out.ctypes.data))
    return out
