"""SNAC codec, host mirror of `class SNAC: Module, AudioCodecModel`
(Sources/MLXAudioCodecs/SNAC/SNACDecoder.swift:11-205).  All arithmetic runs in libmi_speech.so."""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .generation import AudioGenerationError, check


@dataclass
class SNACConfig:
    """SNAC/Config.swift:10-37"""
    sampling_rate: int = 24000
    encoder_dim: int = 48
    encoder_rates: list = field(default_factory=lambda: [2, 4, 8, 8])
    latent_dim: int | None = None
    decoder_dim: int = 1024
    decoder_rates: list = field(default_factory=lambda: [8, 8, 4, 2])
    attn_window_size: int | None = None
    codebook_size: int = 4096
    codebook_dim: int = 8
    vq_strides: list = field(default_factory=lambda: [4, 2, 1])
    noise: bool = True
    depthwise: bool = True

    @classmethod
    def from_dict(cls, d: dict) -> "SNACConfig":
        keys = cls.__dataclass_fields__.keys()
        return cls(**{k: d[k] for k in keys if k in d})

    def to_c(self) -> "_lib.SnacConfigC":
        c = _lib.SnacConfigC()
        c.sampling_rate = self.sampling_rate
        c.latent_dim = self.latent_dim or self.encoder_dim * 2 ** len(self.encoder_rates)
        c.decoder_dim = self.decoder_dim
        c.n_decoder_rates = len(self.decoder_rates)
        for i, r in enumerate(self.decoder_rates):
            c.decoder_rates[i] = r
        c.codebook_size, c.codebook_dim = self.codebook_size, self.codebook_dim
        c.n_codebooks = len(self.vq_strides)
        for i, s in enumerate(self.vq_strides):
            c.vq_strides[i] = s
        c.noise = 1 if self.noise else 0
        c.depthwise = 1 if self.depthwise else 0
        c.attn_window_size = self.attn_window_size or 0
        return c


_NP_DTYPES = {np.dtype(np.float32): _lib.MIS_F32, np.dtype(np.float16): _lib.MIS_F16}


def _tensor_args(arr):
    """numpy / torch tensor -> (keepalive, data pointer, mis dtype, shape)."""
    try:
        import torch
        if isinstance(arr, torch.Tensor):
            t = arr.detach().contiguous()
            if t.dtype == torch.bfloat16:
                return t, t.data_ptr(), _lib.MIS_BF16, tuple(t.shape)
            if t.dtype == torch.float16:
                return t, t.data_ptr(), _lib.MIS_F16, tuple(t.shape)
            t = t.to(torch.float32).contiguous()
            return t, t.data_ptr(), _lib.MIS_F32, tuple(t.shape)
    except ImportError:
        pass
    a = np.ascontiguousarray(arr)
    if a.dtype not in _NP_DTYPES:
        a = a.astype(np.float32)
    return a, a.ctypes.data, _NP_DTYPES[a.dtype], a.shape


class SNAC:
    """AudioCodecModel conformance (AudioCodecModel.swift:15-27): codec_sample_rate, encode_audio (SNACDecoder.swift:120-125, on
    mis_snac_encode; a handle loaded without encoder tensors raises audioEncodingFailed), decode_audio."""

    def __init__(self, config: SNACConfig, device: int = 0, _handle=None):
        self.config = config
        self.device = device
        self._h = _handle
        if self._h is None:
            h = C.c_void_p()
            cfg = config.to_c()
            check(_lib.lib().mis_snac_create(C.byref(cfg), device, C.byref(h)))
            self._h = h
        self.sampling_rate = config.sampling_rate
        self.hop_length = int(np.prod(config.encoder_rates))

    # -- loading (SNACDecoder.swift:133-189) ---------------------------------------------------
    @classmethod
    def from_model_directory(cls, model_dir: str, device: int = 0) -> "SNAC":
        with open(os.path.join(model_dir, "config.json")) as f:
            cfg = SNACConfig.from_dict(json.load(f))
        h = C.c_void_p()
        check(_lib.lib().mis_snac_load(model_dir.encode(), device, C.byref(h)))
        return cls(cfg, device, _handle=h)

    @classmethod
    def from_pretrained(cls, model_repo: str, device: int = 0) -> "SNAC":
        """fromPretrained (SNACDecoder.swift:133-154).  No network here: the repo id must resolve to a
        local directory (HF snapshot layout or a plain path)."""
        if os.path.isdir(model_repo):
            return cls.from_model_directory(model_repo, device)
        raise AudioGenerationError(1, f"model repo {model_repo!r} is not a local directory (no network access)")

    @classmethod
    def from_weights(cls, config: SNACConfig, weights: dict, device: int = 0) -> "SNAC":
        m = cls(config, device)
        for name, arr in weights.items():
            m.set_tensor(name, arr)                  # "encoder.*" / "*.in_proj.*" enable the encode path when present
        m.finalize()
        return m

    def set_tensor(self, name: str, arr):
        keep, ptr, dt, shape = _tensor_args(arr)
        sh = (C.c_int64 * len(shape))(*shape)
        check(_lib.lib().mis_snac_set_tensor(self._h, name.encode(), ptr, dt, sh, len(shape)))

    def finalize(self):
        check(_lib.lib().mis_snac_finalize(self._h))

    def set_noise(self, null_noise_is_zero: bool, seed: int = 0):
        check(_lib.lib().mis_snac_set_noise(self._h, 1 if null_noise_is_zero else 0, seed))

    # -- AudioCodecModel -----------------------------------------------------------------------
    @property
    def codec_sample_rate(self) -> float:
        return float(self.sampling_rate)

    def num_samples(self, t_coarse: int) -> int:
        return int(_lib.lib().mis_snac_num_samples(self._h, t_coarse))

    def noise_lengths(self, t_coarse: int):
        return [int(_lib.lib().mis_snac_noise_len(self._h, i, t_coarse)) for i in range(len(self.config.decoder_rates))]

    def decode(self, codes, noise=None) -> np.ndarray:
        """SNAC.decode (SNACDecoder.swift:127-131): list of int arrays [B, T_i] -> float32 [B, 1, N].
        noise: None (policy of set_noise) or one [B, T_i] array per decoder block."""
        codes = [np.ascontiguousarray(c, dtype=np.int32) for c in codes]
        if len(codes) != len(self.config.vq_strides):
            raise AudioGenerationError(3, f"expected {len(self.config.vq_strides)} code arrays")
        if any(c.ndim != 2 for c in codes):
            raise AudioGenerationError(3, "codes must be [batch, time]")
        B, t_coarse = codes[0].shape
        s0 = self.config.vq_strides[0]
        for c, s in zip(codes, self.config.vq_strides):
            if c.shape != (B, t_coarse * (s0 // s)):
                raise AudioGenerationError(3, "code array lengths inconsistent with vq_strides")
        N = self.num_samples(t_coarse)
        out = np.zeros((B, 1, N), np.float32)
        if B == 0 or t_coarse == 0:
            return out
        cptr = (C.c_void_p * len(codes))(*[c.ctypes.data for c in codes])
        nptr = None
        keep = None
        if noise is not None:
            keep = [np.ascontiguousarray(n, dtype=np.float32) for n in noise]
            lens = self.noise_lengths(t_coarse)
            for n, L in zip(keep, lens):
                if n.shape != (B, L):
                    raise AudioGenerationError(3, f"noise shape {n.shape} != {(B, L)}")
            nptr = (C.c_void_p * len(keep))(*[n.ctypes.data for n in keep])
        check(_lib.lib().mis_snac_decode(self._h, cptr, B, t_coarse, nptr, out.ctypes.data))
        return out

    def decode_audio(self, codes) -> np.ndarray:           # decodeAudio, SNACDecoder.swift:201-203
        return self.decode(codes)

    def padded_length(self, n_samples: int) -> int:
        return int(_lib.lib().mis_snac_padded_length(self._h, n_samples))

    def encode(self, audio, return_latent: bool = False):
        """SNAC.encode (SNACDecoder.swift:120-125): audio [B, samples] (or [samples]) -> list of int32 codes [B, T_i]."""
        a = np.ascontiguousarray(audio, dtype=np.float32)
        if a.ndim == 1:
            a = a[None]
        if a.ndim == 3 and a.shape[1] == 1:
            a = np.ascontiguousarray(a[:, 0])
        B, n = a.shape
        Tl = self.padded_length(n) // self.hop_length
        outs = [np.zeros((B, Tl // s), np.int32) for s in self.config.vq_strides]
        ptrs = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        latent = self.config.latent_dim or self.config.encoder_dim * 2 ** len(self.config.encoder_rates)
        z = np.zeros((B, latent, Tl), np.float32) if return_latent else None
        check(_lib.lib().mis_snac_encode(self._h, a.ctypes.data, B, n, ptrs, z.ctypes.data if return_latent else None))
        return (outs, z) if return_latent else outs

    def encode_audio(self, waveform):                      # encodeAudio, SNACDecoder.swift:197-199
        return self.encode(waveform)

    def debug_tap(self, name: str, batch: int) -> np.ndarray:
        """Intermediate [batch, C, T] of the LAST decode: "zq", "stem_dw", "stem_pw", "block<i>"."""
        cap = 1 << 26
        buf = np.zeros(cap, np.float32)
        ch, ln = C.c_int32(), C.c_int64()
        check(_lib.lib().mis_snac_debug_tap(self._h, name.encode(), buf.ctypes.data, cap, C.byref(ch), C.byref(ln)))
        return buf[: batch * ch.value * ln.value].reshape(batch, ch.value, ln.value).copy()

    def close(self):
        if self._h is not None:
            _lib.lib().mis_snac_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class DescriptDACConfig:
    """DescriptDACConfig (Sources/MLXAudioCodecs/Descript/DescriptDACConfig.swift:3-34)"""
    encoder_dim: int = 64
    encoder_rates: tuple = (2, 4, 5, 8)
    latent_dim: int | None = None
    decoder_dim: int = 1536
    decoder_rates: tuple = (8, 5, 4, 2)
    n_codebooks: int = 12
    codebook_size: int = 1024
    codebook_dim: int = 8
    sample_rate: int = 16000

    @property
    def resolved_latent(self) -> int:
        return self.latent_dim or self.encoder_dim * 2 ** len(self.encoder_rates)

    def to_c(self) -> "_lib.DacConfigC":
        return _lib.DacConfigC(self.resolved_latent, self.decoder_dim, len(self.decoder_rates), (C.c_int32 * 8)(*self.decoder_rates),
                               self.n_codebooks, self.codebook_size, self.codebook_dim, self.sample_rate)


class DescriptDAC:
    """class DescriptDAC (Descript/DescriptDAC.swift:172-245,334-352): preprocess / encode / encode_audio / decode_from_codes /
    decode_audio.  The encoder needs the checkpoint's encoder.* and in_proj tensors (error 5 otherwise)."""

    def __init__(self, config: DescriptDACConfig, device: int = 0):
        self.config = config
        self._h = C.c_void_p()
        cc = config.to_c()
        check(_lib.lib().mis_dac_create(C.byref(cc), device, C.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().mis_dac_destroy(h)

    @classmethod
    def from_weights(cls, config, weights: dict, device: int = 0) -> "DescriptDAC":
        m = cls(config, device)
        for k, v in weights.items():
            keep, ptr, dt, shape = _tensor_args(v)
            sh = (C.c_int64 * len(shape))(*shape)
            check(_lib.lib().mis_dac_set_tensor(m._h, k.encode(), ptr, dt, sh, len(shape)))
        check(_lib.lib().mis_dac_finalize(m._h))
        return m

    @property
    def codec_sample_rate(self) -> float:
        return float(self.config.sample_rate)

    def num_samples(self, n_frames: int) -> int:
        return int(_lib.lib().mis_dac_num_samples(self._h, n_frames))

    def decode_from_codes(self, codes) -> np.ndarray:
        """codes int [B, n_codebooks, T] -> waveform float32 [B, num_samples(T)]"""
        cd = np.ascontiguousarray(codes, dtype=np.int32)
        B, ncb, T = cd.shape
        if ncb != self.config.n_codebooks:
            raise AudioGenerationError(3, "wrong number of codebooks")
        out = np.zeros((B, self.num_samples(T)), np.float32)
        check(_lib.lib().mis_dac_decode_codes(self._h, cd.ctypes.data, B, T, out.ctypes.data))
        return out

    def debug_tap(self, codes, block: int) -> np.ndarray:
        cd = np.ascontiguousarray(codes, dtype=np.int32)
        B, ncb, T = cd.shape
        cap = B * self.config.decoder_dim * self.num_samples(T)
        buf = np.zeros(cap, np.float32)
        ch = C.c_int32(); ln = C.c_int64()
        check(_lib.lib().mis_dac_debug_tap(self._h, cd.ctypes.data, B, T, block, buf.ctypes.data, cap, C.byref(ch), C.byref(ln)))
        return buf[: B * ch.value * ln.value].reshape(B, ch.value, ln.value).copy()

    @property
    def hop_length(self) -> int:
        return int(np.prod(self.config.encoder_rates))

    def preprocess(self, audio, sample_rate: int | None = None) -> np.ndarray:
        """DescriptDAC.preprocess (:216-228): right-pad [B, n] to a multiple of the hop length"""
        if sample_rate is not None and sample_rate != self.config.sample_rate:
            raise AudioGenerationError(3, f"Sample rate mismatch: {sample_rate} != {self.config.sample_rate}")
        a = np.ascontiguousarray(audio, dtype=np.float32)
        pad = -a.shape[1] % self.hop_length
        return np.pad(a, ((0, 0), (0, pad))) if pad else a

    def encode(self, audio, n_quantizers: int | None = None, return_latent: bool = False):
        """DescriptDAC.encode (:230-233) on preprocess()ed audio [B, n]: codes int32 [B, nq, n / hop] (+ the encoder output
        [B, latent, T] before the RVQ when return_latent).  Un-padded input is padded as preprocess() does."""
        a = np.ascontiguousarray(audio, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] < 1:
            raise AudioGenerationError(3, "audio must be [batch, samples]")
        B, n = a.shape
        nq = self.config.n_codebooks if n_quantizers is None else int(n_quantizers)
        if not 1 <= nq <= self.config.n_codebooks:
            raise AudioGenerationError(3, "n_quantizers out of range")
        T = int(_lib.lib().mis_dac_padded_length(self._h, n)) // self.hop_length
        if T < 1:
            raise AudioGenerationError(5, "this DAC model was loaded without encoder weights")
        codes = np.zeros((B, nq, T), np.int32)
        z = np.zeros((B, self.config.resolved_latent, T), np.float32) if return_latent else None
        check(_lib.lib().mis_dac_encode(self._h, a.ctypes.data, B, n, nq, codes.ctypes.data, z.ctypes.data if return_latent else None))
        return (codes, z) if return_latent else codes

    def encode_audio(self, waveform) -> dict:
        """AudioCodecModel.encodeAudio (:340-345): {"codes", "original_length"}"""
        w = np.ascontiguousarray(waveform, dtype=np.float32)
        return {"codes": self.encode(self.preprocess(w, self.config.sample_rate)), "original_length": int(w.shape[1])}

    def decode_audio(self, encoded: dict) -> np.ndarray:
        """AudioCodecModel.decodeAudio (:347-350): decode and trim to the original length"""
        return self.decode_from_codes(encoded["codes"])[:, : encoded["original_length"]]


@dataclass
class EncodecConfig:
    """EncodecConfig (Sources/MLXAudioCodecs/Encodec/EncodecConfig.swift:64-89)"""
    audio_channels: int = 1
    num_filters: int = 32
    kernel_size: int = 7
    num_residual_layers: int = 1
    dilation_growth_rate: int = 2
    codebook_size: int = 1024
    codebook_dim: int = 128
    hidden_size: int = 128
    num_lstm_layers: int = 2
    residual_kernel_size: int = 3
    use_causal_conv: bool = True
    pad_mode: str = "reflect"
    norm_type: str = "weight_norm"
    last_kernel_size: int = 7
    trim_right_ratio: float = 1.0
    compress: int = 2
    upsampling_ratios: tuple = (8, 5, 4, 2)
    target_bandwidths: tuple = (1.5, 3.0, 6.0, 12.0, 24.0)
    sampling_rate: int = 24000
    chunk_length_s: float | None = None
    overlap: float | None = None
    use_conv_shortcut: bool = True

    @property
    def hop_length(self) -> int:
        return int(np.prod(self.upsampling_ratios))

    @property
    def num_quantizers(self) -> int:              # EncodecQuantization.swift:60-64
        import math
        frame_rate = int(math.ceil(self.sampling_rate / self.hop_length))
        return int(1000 * max(self.target_bandwidths) / (frame_rate * 10))

    @property
    def chunk_length(self):                       # Encodec.swift:196-201
        return None if self.chunk_length_s is None else int(self.chunk_length_s * self.sampling_rate)

    @property
    def chunk_stride(self):                       # Encodec.swift:203-208
        if self.chunk_length_s is None or self.overlap is None:
            return None
        return max(1, int((1.0 - self.overlap) * self.chunk_length))

    def to_c(self) -> "_lib.EncodecConfigC":
        if self.norm_type not in ("weight_norm", "time_group_norm"):
            raise AudioGenerationError(3, f"unknown Encodec norm_type {self.norm_type!r}")
        return _lib.EncodecConfigC(self.audio_channels, self.num_filters, self.kernel_size, self.num_residual_layers,
                                   self.dilation_growth_rate, self.codebook_size, self.codebook_dim, self.hidden_size,
                                   self.num_lstm_layers, self.residual_kernel_size, 1 if self.use_causal_conv else 0,
                                   1 if self.pad_mode == "reflect" else 0, self.last_kernel_size, self.compress,
                                   1 if self.use_conv_shortcut else 0, float(self.trim_right_ratio), len(self.upsampling_ratios),
                                   (C.c_int32 * 8)(*self.upsampling_ratios), self.num_quantizers, self.sampling_rate,
                                   1 if self.norm_type == "time_group_norm" else 0)


class Encodec:
    """Decode side of class Encodec (Encodec/Encodec.swift:179-398): decode(audio_codes, audio_scales); the 24 kHz (mono, causal,
    plain convs) and the 48 kHz (stereo, non-causal, GroupNorm) model families."""

    def __init__(self, config: EncodecConfig, device: int = 0):
        self.config = config
        self._h = C.c_void_p()
        cc = config.to_c()
        check(_lib.lib().mis_encodec_create(C.byref(cc), device, C.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().mis_encodec_destroy(h)

    @classmethod
    def from_weights(cls, config, weights: dict, device: int = 0) -> "Encodec":
        m = cls(config, device)
        for k, v in weights.items():
            keep, ptr, dt, shape = _tensor_args(v)
            sh = (C.c_int64 * len(shape))(*shape)
            check(_lib.lib().mis_encodec_set_tensor(m._h, k.encode(), ptr, dt, sh, len(shape)))
        check(_lib.lib().mis_encodec_finalize(m._h))
        return m

    @property
    def codec_sample_rate(self) -> float:
        return float(self.config.sampling_rate)

    def decode_frame(self, codes, scale=None) -> np.ndarray:
        """decodeFrame (Encodec.swift:295-302): codes int [B, n_q, T] -> [B, T * hop] (mono) or [B, channels, T * hop]"""
        cd = np.ascontiguousarray(codes, dtype=np.int32)
        B, nq, T = cd.shape
        ch = self.config.audio_channels
        out = np.zeros((B, T * self.config.hop_length) if ch == 1 else (B, ch, T * self.config.hop_length), np.float32)
        sc = None if scale is None else np.ascontiguousarray(np.broadcast_to(np.asarray(scale, np.float32).reshape(-1), (B,)))
        check(_lib.lib().mis_encodec_decode_frame(self._h, cd.ctypes.data, B, nq, T, None if sc is None else sc.ctypes.data,
                                                  out.ctypes.data))
        return out

    @staticmethod
    def linear_overlap_add(frames, hop_stride: int) -> np.ndarray:
        """Encodec.linearOverlapAdd (Encodec.swift:304-355); host arithmetic in the reference as well."""
        L = frames[0].shape[1]
        total = hop_stride * (len(frames) - 1) + frames[-1].shape[1]
        tv = (np.arange(L, dtype=np.float32) + np.float32(1)) / np.float32(L + 1)
        wv = (np.float32(0.5) - np.abs(tv - np.float32(0.5))).astype(np.float32)
        out = np.zeros((frames[0].shape[0], total), np.float32)
        sw = np.zeros(total, np.float32)
        off = 0
        for f in frames:
            n = f.shape[1]
            out[:, off:off + n] += wv[:n] * f
            sw[off:off + n] += wv[:n]
            off += hop_stride
        nz = sw != 0
        out[:, nz] /= sw[nz]
        return out

    def _overlap_add(self, frames, hop_stride: int) -> np.ndarray:
        if frames[0].ndim == 2:
            return self.linear_overlap_add(frames, hop_stride)
        B, ch = frames[0].shape[:2]                          # stereo: the time axis is the last one; channels ride along as rows
        out = self.linear_overlap_add([f.reshape(B * ch, f.shape[-1]) for f in frames], hop_stride)
        return out.reshape(B, ch, -1)

    def decode(self, audio_codes, audio_scales=None, padding_mask=None) -> np.ndarray:
        """decode (Encodec.swift:357-398): audio_codes [n_chunks, B, n_q, frames] -> [B, samples]"""
        ac = np.asarray(audio_codes)
        scales = audio_scales if audio_scales is not None else [None] * ac.shape[0]
        if self.config.chunk_length is None:
            if ac.shape[0] != 1:
                raise AudioGenerationError(3, f"Expected one frame, got {ac.shape[0]}")
            out = self.decode_frame(ac[0], scales[0])
        else:
            out = self._overlap_add([self.decode_frame(ac[i], scales[i]) for i in range(ac.shape[0])], self.config.chunk_stride or 1)
        if padding_mask is not None and np.asarray(padding_mask).shape[1] < out.shape[-1]:
            out = out[..., : np.asarray(padding_mask).shape[1]]
        return out

    def debug_tap(self, codes, stage: int) -> np.ndarray:
        cd = np.ascontiguousarray(codes, dtype=np.int32)
        B, nq, T = cd.shape
        cap = B * max(self.config.num_filters << len(self.config.upsampling_ratios), 4) * (T + 8) * self.config.hop_length
        buf = np.zeros(cap, np.float32)
        ch = C.c_int32(); ln = C.c_int64()
        check(_lib.lib().mis_encodec_debug_tap(self._h, cd.ctypes.data, B, nq, T, stage, buf.ctypes.data, cap, C.byref(ch), C.byref(ln)))
        return buf[: B * ch.value * ln.value].reshape(B, ch.value, ln.value).copy()
