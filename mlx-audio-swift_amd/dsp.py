"""Mel / STFT front end, host mirror of Sources/MLXAudioCore/DSP.swift (computeMelSpectrogram :230-273) and
WhisperAudio (Sources/MLXAudioSTT/Models/Whisper/WhisperAudio.swift: logMelSpectrogram :38-79, encoderFeatures
:83-87).  All arithmetic runs in libmi_speech.so (csrc/mel.hip)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .generation import AudioGenerationError, check

WHISPER_SAMPLE_RATE, WHISPER_N_FFT, WHISPER_HOP, WHISPER_CHUNK_SAMPLES, WHISPER_N_FRAMES = 16000, 400, 160, 480000, 3000


def _mel(cfg: "_lib.MelConfigC", audio: np.ndarray, device: int) -> np.ndarray:
    a = np.ascontiguousarray(audio, dtype=np.float32)
    squeeze = a.ndim == 1
    if squeeze:
        a = a[None]
    if a.ndim != 2:
        raise AudioGenerationError(3, "audio must be [samples] or [batch, samples]")
    B, n = a.shape
    frames = int(_lib.lib().mis_mel_num_frames(C.byref(cfg), n))
    out = np.zeros((B, frames, cfg.n_mels), np.float32)
    nf = C.c_int64()
    check(_lib.lib().mis_mel_spectrogram(device, C.byref(cfg), a.ctypes.data if n else None, B, n,
                                         out.ctypes.data if frames else None, C.byref(nf)))
    return out[0] if squeeze else out


def compute_mel_spectrogram(audio, sample_rate: int, n_fft: int, hop_length: int, n_mels: int, device: int = 0):
    """computeMelSpectrogram (DSP.swift:230-273): symmetric Hann, HTK mel scale with Slaney norm -> [frames, n_mels]."""
    return _mel(_lib.MelConfigC(sample_rate, n_fft, hop_length, n_mels, 1, 0, 1, 0), audio, device)


def log_mel_spectrogram(audio, n_mels: int, device: int = 0) -> np.ndarray:
    """WhisperAudio.logMelSpectrogram (:38-79) -> [n_mels, n_frames] (no 30 s padding)."""
    m = _mel(_lib.MelConfigC(WHISPER_SAMPLE_RATE, WHISPER_N_FFT, WHISPER_HOP, n_mels, 0, 1, 1, 1),
             np.asarray(audio, np.float32).reshape(-1), device)
    return np.ascontiguousarray(m.T)


def whisper_encoder_features(audio, n_mels: int, lens=None, device: int = 0) -> np.ndarray:
    """WhisperAudio.encoderFeatures (:83-87) for a batch: [B, samples] (or [samples]) -> [B, 3000, n_mels]."""
    a = np.ascontiguousarray(audio, dtype=np.float32)
    if a.ndim == 1:
        a = a[None]
    B, stride = a.shape
    out = np.zeros((B, WHISPER_N_FRAMES, n_mels), np.float32)
    lp = None
    if lens is not None:
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        lp = lens.ctypes.data
    check(_lib.lib().mis_whisper_encoder_features(device, a.ctypes.data if stride else None, lp, B, stride, n_mels,
                                                  out.ctypes.data))
    return out


class IncrementalMelSpectrogram:
    """class IncrementalMelSpectrogram (Sources/MLXAudioSTT/Streaming/IncrementalMelSpectrogram.swift:17-215):
    process(samples) -> [new_frames, n_mels] or None; flush(); reset(); total_frames."""

    def __init__(self, sample_rate: int = 16000, n_fft: int = 400, hop_length: int = 160, n_mels: int = 128, device: int = 0):
        self.n_fft, self.hop_length, self.n_mels = n_fft, hop_length, n_mels
        self._h = C.c_void_p()
        check(_lib.lib().mis_mel_stream_create(device, sample_rate, n_fft, hop_length, n_mels, C.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().mis_mel_stream_destroy(h)

    @property
    def total_frames(self) -> int:
        return int(_lib.lib().mis_mel_stream_total_frames(self._h))

    def process(self, samples):
        a = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1)
        if a.size == 0:
            return None
        cap = (a.size + self.n_fft) // self.hop_length + 2
        out = np.zeros((cap, self.n_mels), np.float32)
        n = C.c_int64()
        check(_lib.lib().mis_mel_stream_process(self._h, a.ctypes.data, a.size, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].copy() if n.value else None

    def flush(self):
        cap = (2 * self.n_fft) // self.hop_length + 4
        out = np.zeros((cap, self.n_mels), np.float32)
        n = C.c_int64()
        check(_lib.lib().mis_mel_stream_flush(self._h, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].copy() if n.value else None

    def reset(self):
        check(_lib.lib().mis_mel_stream_reset(self._h))
