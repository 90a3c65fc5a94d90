"""Log-mel front end: CPU restatement of the reference.  Test infrastructure only.

Follows:
  * WhisperAudio.padOrTrimToWindow / logMelSpectrogram / encoderFeatures / reflectPad
        Sources/MLXAudioSTT/Models/Whisper/WhisperAudio.swift:7-13, 38-79, 83-87, 89-112
  * melFilters (HTK / Slaney scale, Slaney norm)     Sources/MLXAudioCore/DSP.swift:76-168
  * hanningWindow (symmetric, N-1)                    DSP.swift:15-22
  * stft / computeMelSpectrogram (generic DSP path)   DSP.swift:181-227, 230-273
Third-party semantics restated [3P mlx-swift]: MLXFFT.rfft = unnormalised forward DFT, bins 0..n/2;
asStrided frames = overlapping windows of the padded signal.  float32 arithmetic as in MLX (numpy's
pocketfft runs rfft natively in float32).  Pinned against HF transformers' WhisperFeatureExtractor
(an independent numpy implementation of OpenAI's front end) in tests/test_oracle_mel.py, and against the
known answers the reference's own tests hold: feature shape [1,3000,80] for 5 s of zeros
(Tests/MLXAudioSTTTests.swift:4416-4422) with value (log10(1e-10)+4)/4 = -1.5 everywhere, and the
Hamming spot values of Tests/MLXAudioCodecsTests.swift:117-131.
"""
from __future__ import annotations

import numpy as np

F = np.float32
SAMPLE_RATE, N_FFT, HOP, CHUNK_SAMPLES, N_FRAMES = 16000, 400, 160, 480000, 3000   # WhisperConfig.swift:188-193


def hanning_window(size: int) -> np.ndarray:            # DSP.swift:15-22 (symmetric)
    n = np.arange(size, dtype=F)
    return (F(0.5) * (F(1.0) - np.cos(F(2.0) * F(np.pi) * n / F(size - 1)))).astype(F)


def hamming_window(size: int, periodic: bool = True) -> np.ndarray:     # DSP.swift:25-43
    if size <= 0:
        return np.zeros(0, F)
    if size == 1:
        return np.ones(1, F)
    eff = size + 1 if periodic else size
    n = np.arange(eff, dtype=F)
    v = (F(0.54) - F(0.46) * np.cos(F(2.0) * F(np.pi) * n / F(eff - 1))).astype(F)
    return v[:size]


def whisper_window() -> np.ndarray:                     # WhisperAudio.swift:42-43 (periodic Hann)
    n = np.arange(N_FFT, dtype=F)
    return (F(0.5) * (F(1.0) - np.cos((F(2.0) * F(np.pi) * n) / F(N_FFT)))).astype(F)


def mel_filters(sample_rate: int, n_fft: int, n_mels: int, f_min: float = 0.0, f_max: float | None = None,
                norm: str | None = "slaney", mel_scale: str = "htk") -> np.ndarray:
    """DSP.swift:76-168, float32 scalar arithmetic.  Returns [n_freqs, n_mels]."""
    f_min = F(f_min)
    f_max = F(sample_rate) / F(2.0) if f_max is None else F(f_max)
    n_freqs = n_fft // 2 + 1
    all_freqs = (np.arange(n_freqs, dtype=F) * F(sample_rate) / F(n_fft)).astype(F)
    if mel_scale == "htk":
        hz_to_mel = lambda f: F(2595.0) * np.log10(F(1.0) + F(f) / F(700.0)).astype(F)
        mel_to_hz = lambda m: F(700.0) * (np.power(F(10.0), F(m) / F(2595.0)).astype(F) - F(1.0))
    else:
        f_sp = F(200.0) / F(3.0)
        min_log_hz = F(1000.0)
        min_log_mel = (min_log_hz - f_min) / f_sp
        log_step = F(np.log(F(6.4))) / F(27.0)

        def hz_to_mel(f):
            f = F(f)
            return (f - f_min) / f_sp if f < min_log_hz else min_log_mel + F(np.log(f / min_log_hz)) / log_step

        def mel_to_hz(m):
            m = F(m)
            return f_min + f_sp * m if m < min_log_mel else min_log_hz * F(np.exp(log_step * (m - min_log_mel)))
    m_min, m_max = F(hz_to_mel(f_min)), F(hz_to_mel(f_max))
    m_pts = [F(m_min + F(i) * (m_max - m_min) / F(n_mels + 1)) for i in range(n_mels + 2)]
    f_pts = [F(mel_to_hz(m)) for m in m_pts]
    fb = np.zeros((n_freqs, n_mels), F)
    for j in range(n_mels):
        low, center, high = f_pts[j], f_pts[j + 1], f_pts[j + 2]
        for i in range(n_freqs):
            f = all_freqs[i]
            if low <= f < center:
                fb[i, j] = (f - low) / (center - low)
            elif center <= f <= high:
                fb[i, j] = (high - f) / (high - center)
        if norm == "slaney":
            fb[:, j] *= F(2.0) / (high - low)
    return fb


def reflect_pad(audio: np.ndarray, pad: int) -> np.ndarray:      # WhisperAudio.swift:89-112
    audio = np.asarray(audio, F)
    n = audio.shape[0]
    if pad <= 0:
        return audio
    if n <= 1:
        return np.concatenate([np.zeros(pad, F), audio, np.zeros(pad, F)])
    lc = min(pad, n - 1)
    left = audio[1:lc + 1][::-1]
    right = audio[n - 1 - lc:n - 1][::-1]
    return np.concatenate([np.zeros(pad - lc, F), left, audio, right, np.zeros(pad - lc, F)])


def pad_or_trim(audio: np.ndarray, target: int = CHUNK_SAMPLES) -> np.ndarray:      # WhisperAudio.swift:7-13
    audio = np.asarray(audio, F)
    if audio.shape[0] >= target:
        return audio[:target]
    return np.concatenate([audio, np.zeros(target - audio.shape[0], F)])


def log_mel_spectrogram(audio: np.ndarray, n_mels: int) -> np.ndarray:
    """WhisperAudio.logMelSpectrogram (:38-79) -> [n_mels, n_frames] float32."""
    padded = reflect_pad(np.asarray(audio, F).reshape(-1), N_FFT // 2)
    n_frames = 1 + (padded.shape[0] - N_FFT) // HOP if padded.shape[0] >= N_FFT else 0
    if n_frames <= 0:
        return np.zeros((n_mels, 0), F)
    idx = np.arange(n_frames)[:, None] * HOP + np.arange(N_FFT)[None, :]
    windowed = (padded[idx] * whisper_window()[None, :]).astype(F)
    spec = np.fft.rfft(windowed, axis=-1)
    mag = (np.abs(spec).astype(F) ** 2).astype(F)
    mag = mag[:-1].T                                              # drop last frame (:65-67), [201, frames]
    if mag.shape[1] == 0:
        return np.zeros((n_mels, 0), F)
    filters = mel_filters(SAMPLE_RATE, N_FFT, n_mels, 0.0, SAMPLE_RATE / 2.0, "slaney", "slaney")
    mel = (filters.T.astype(F) @ mag).astype(F)
    mel = np.maximum(mel, F(1e-10))
    log_spec = np.log10(mel).astype(F)
    log_spec = np.maximum(log_spec, log_spec.max() - F(8.0))
    return ((log_spec + F(4.0)) / F(4.0)).astype(F)


def encoder_features(audio: np.ndarray, n_mels: int) -> np.ndarray:      # WhisperAudio.swift:83-87 -> [1, frames, mels]
    return log_mel_spectrogram(pad_or_trim(audio), n_mels).T[None]


def compute_mel_spectrogram(audio: np.ndarray, sample_rate: int, n_fft: int, hop: int, n_mels: int) -> np.ndarray:
    """Generic DSP path, DSP.swift:181-273: symmetric Hann, reflect pad, HTK scale + Slaney norm, all frames
    kept, [frames, mels]."""
    audio = np.asarray(audio, F)
    n, pad = audio.shape[0], n_fft // 2
    prefix = audio[1:min(pad + 1, n)][::-1]
    suffix = audio[max(0, n - pad - 1):max(1, n - 1)][::-1]
    padded = np.concatenate([prefix, audio, suffix])
    n_frames = 1 + (padded.shape[0] - n_fft) // hop
    idx = np.arange(n_frames)[:, None] * hop + np.arange(n_fft)[None, :]
    spec = np.fft.rfft((padded[idx] * hanning_window(n_fft)[None, :]).astype(F), axis=1)
    mag = (np.abs(spec).astype(F) ** 2).astype(F)
    mel = (mag @ mel_filters(sample_rate, n_fft, n_mels, norm="slaney")).astype(F)
    mel = np.log10(np.maximum(mel, F(1e-10))).astype(F)
    mel = np.maximum(mel, mel.max() - F(8.0))
    return ((mel + F(4.0)) / F(4.0)).astype(F)


class IncrementalMelOracle:
    """IncrementalMelSpectrogram (Sources/MLXAudioSTT/Streaming/IncrementalMelSpectrogram.swift:17-215): overlap-save framing
    with a reflected prefix on the first chunk, symmetric Hann window, HTK-scale / Slaney-norm filters, log10 with a RUNNING
    maximum (grows monotonically over the session), (x + 4) / 4."""

    def __init__(self, sample_rate=16000, n_fft=400, hop_length=160, n_mels=128):
        self.n_fft, self.hop, self.n_mels = n_fft, hop_length, n_mels
        self.overlap_size = n_fft - hop_length
        self.window = hanning_window(n_fft)
        self.filters = mel_filters(sample_rate, n_fft, n_mels, norm="slaney", mel_scale="htk")
        self.reset()

    def reset(self):
        self.overlap = np.zeros(0, F)
        self.first = True
        self.running_max = -np.inf
        self.total_frames = 0

    def _frames(self, signal, n):
        idx = np.arange(n)[:, None] * self.hop + np.arange(self.n_fft)[None, :]
        spec = np.fft.rfft((signal[idx] * self.window).astype(F), axis=1)
        power = (np.abs(spec).astype(F) ** 2).astype(F)
        mel = np.log10(np.maximum((power @ self.filters).astype(F), F(1e-10))).astype(F)
        self.running_max = max(self.running_max, float(mel.max()))
        mel = np.maximum(mel, F(self.running_max - 8.0))
        self.total_frames += n
        return ((mel + F(4.0)) / F(4.0)).astype(F)

    def process(self, samples):
        samples = np.asarray(samples, F)
        if samples.size == 0:
            return None
        if self.first:                                         # :77-99
            pad = self.n_fft // 2
            prefix = np.zeros(0, F)
            if samples.size > 1:
                rl = min(pad, samples.size - 1)
                if rl > 0:
                    prefix = samples[1:rl + 1][::-1]
            if prefix.size == 0:
                prefix = np.full(pad, samples[0], F)
            else:
                while prefix.size < pad:
                    prefix = np.concatenate([prefix, prefix[: pad - prefix.size]])
            signal = np.concatenate([prefix, samples]).astype(F)
            self.first = False
        else:
            signal = np.concatenate([self.overlap, samples]).astype(F)
        n = max(0, (signal.size - self.n_fft) // self.hop + 1)
        if n <= 0:
            self.overlap = signal
            return None
        consumed = (n - 1) * self.hop + self.n_fft
        self.overlap = signal[consumed - self.overlap_size:] if consumed < signal.size else signal[signal.size - self.overlap_size:]
        return self._frames(signal, n)

    def flush(self):                                          # :150-200
        if self.overlap.size == 0:
            return None
        signal = self.overlap
        if signal.size < self.n_fft:
            signal = np.concatenate([signal, np.zeros(self.n_fft - signal.size, F)])
        pad = self.n_fft // 2
        rl = min(pad, signal.size - 1)
        signal = np.concatenate([signal, signal[signal.size - 1 - rl: signal.size - 1][::-1]]).astype(F)
        self.overlap = np.zeros(0, F)
        n = max(0, (signal.size - self.n_fft) // self.hop + 1)
        return self._frames(signal, n) if n > 0 else None
