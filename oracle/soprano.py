"""Soprano decoder (hidden states -> waveform): CPU restatement of the reference.  Test infrastructure only.

Follows:
  * interpolate1d (align_corners=true)        Sources/MLXAudioTTS/Models/Soprano/SopranoDecoder.swift:22-80
  * ISTFTHead (Linear -> exp/clip mag, phase -> irfft -> windowed overlap-add normalised by the window SUM,
    trimmed by n_fft/2 at both ends)           SopranoDecoder.swift:87-218
  * SopranoDecoder                             SopranoDecoder.swift:225-284   interp x upscale -> VocosBackbone -> head
  * ConvNeXtBlock / VocosBackbone              Sources/MLXAudioCodecs/Vocos/VocosBackbone.swift:18-204
  * streamGenerate sampling quirks             Sources/MLXAudioTTS/Models/Soprano/Soprano.swift:801-901, 996-1060
Decoder weights are float32 in the reference (Soprano.swift:332-339).  Third-party semantics restated [3P]:
MLXNN.Conv1d NLC with weight [out, k, in/groups]; LayerNorm eps 1e-6 over the channel axis; gelu exact erf;
MLXFFT.irfft(n = 2*(bins-1)) ignores the imaginary part of the DC and Nyquist bins.
Layout here is [B, L, C] like the reference's MLX convention."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

F = np.float32


@dataclass
class SopranoDecoderConfig:          # SopranoConfig.swift:158-167 (Soprano-1.1 defaults)
    hidden_size: int = 512
    decoder_num_layers: int = 8
    decoder_dim: int = 768
    decoder_intermediate_dim: int = 2304
    hop_length: int = 512
    n_fft: int = 2048
    upscale: int = 4
    input_kernel: int = 1
    dw_kernel: int = 3
    token_size: int = 2048


TINY = SopranoDecoderConfig(hidden_size=64, decoder_num_layers=2, decoder_dim=96, decoder_intermediate_dim=160,
                            hop_length=32, n_fft=128, upscale=4, input_kernel=3, dw_kernel=3, token_size=128)


def interpolate1d(x: np.ndarray, size: int) -> np.ndarray:
    """x [B, C, L] -> [B, C, size], align_corners=True (SopranoDecoder.swift:22-80)."""
    B, C, L = x.shape
    if size < 1 or L < 1 or size == L:
        return x
    if L == 1:
        return np.broadcast_to(x, (B, C, size)).copy()
    pos = (np.arange(size, dtype=F) * (F(L - 1) / F(size - 1))).astype(F)
    lo = np.floor(pos).astype(np.int32)
    hi = np.minimum(lo + 1, L - 1)
    frac = (pos - lo.astype(F)).astype(F)
    return (x[:, :, lo] * (F(1.0) - frac) + x[:, :, hi] * frac).astype(F)


def _layer_norm(x, w, b, eps=1e-6):
    m = x.mean(-1, keepdims=True, dtype=np.float32)
    v = ((x - m) ** 2).mean(-1, keepdims=True, dtype=np.float32)
    return ((x - m) / np.sqrt(v + F(eps)) * w + b).astype(F)


def _gelu(x):
    from scipy.special import erf
    return (F(0.5) * x * (F(1.0) + erf(x / F(math.sqrt(2.0))).astype(F))).astype(F)


def _conv1d_nlc(x, w, b, groups=1):
    """MLXNN.Conv1d, padding k/2.  x [B, L, Cin]; w [Cout, k, Cin/groups]."""
    B, L, Cin = x.shape
    Cout, k, Cg = w.shape
    pad = k // 2
    xp = np.zeros((B, L + 2 * pad, Cin), F)
    xp[:, pad:pad + L] = x
    y = np.zeros((B, L, Cout), F)
    for j in range(k):
        xs = xp[:, j:j + L]
        if groups == 1:
            y += xs @ w[:, j, :].T
        else:                                   # depthwise
            y += xs * w[:, j, 0][None, None, :]
    return (y + b).astype(F)


class SopranoDecoderOracle:
    def __init__(self, cfg: SopranoDecoderConfig, weights: dict):
        self.cfg = cfg
        self.w = {k: np.asarray(v, F) for k, v in weights.items()}

    def backbone(self, x):                      # VocosBackbone.swift:164-203
        w, p = self.w, "decoder.decoder"
        h = _conv1d_nlc(x, w[p + ".embed.weight"], w[p + ".embed.bias"])
        h = _layer_norm(h, w[p + ".norm.weight"], w[p + ".norm.bias"])
        for i in range(self.cfg.decoder_num_layers):
            q = f"{p}.convnext.{i}"
            r = h
            t = _conv1d_nlc(h, w[q + ".dwconv.weight"], w[q + ".dwconv.bias"], groups=h.shape[-1])
            t = _layer_norm(t, w[q + ".norm.weight"], w[q + ".norm.bias"])
            t = (t @ w[q + ".pwconv1.weight"].T + w[q + ".pwconv1.bias"]).astype(F)
            t = _gelu(t)
            t = (t @ w[q + ".pwconv2.weight"].T + w[q + ".pwconv2.bias"]).astype(F)
            t = (w[q + ".gamma"] * t).astype(F)
            h = (r + t).astype(F)
        return _layer_norm(h, w[p + ".final_layer_norm.weight"], w[p + ".final_layer_norm.bias"])

    def head(self, x):                          # ISTFTHead, SopranoDecoder.swift:101-206
        cfg, w = self.cfg, self.w
        n_fft, hop = cfg.n_fft, cfg.hop_length
        h = (x @ w["decoder.head.out.weight"].T + w["decoder.head.out.bias"]).astype(F)      # [B, L, n_fft+2]
        h = np.swapaxes(h, 1, 2)
        half = (n_fft + 2) // 2
        mag = np.minimum(np.exp(h[:, :half]), F(1e2)).astype(F)
        ph = h[:, half:]
        re, im = (mag * np.cos(ph)).astype(F), (mag * np.sin(ph)).astype(F)
        n = np.arange(n_fft, dtype=F)
        window = (F(0.5) - F(0.5) * np.cos(F(2.0) * (F(np.pi) / F(n_fft - 1)) * n)).astype(F)       # :209-217
        outs = []
        for b in range(x.shape[0]):
            frames = np.fft.irfft((re[b] + 1j * im[b]).astype(np.complex64), n=n_fft, axis=0).astype(F)   # [n_fft, L]
            wf = (frames.T * window).astype(F)
            L = wf.shape[0]
            out_len = (L - 1) * hop + n_fft
            audio = np.zeros(out_len, F)
            wsum = np.zeros(out_len, F)
            for i in range(L):
                audio[i * hop:i * hop + n_fft] += wf[i]
                wsum[i * hop:i * hop + n_fft] += window
            nz = wsum != 0
            audio[nz] = audio[nz] / wsum[nz]
            t0, t1 = n_fft // 2, out_len - n_fft // 2
            outs.append(audio[t0:t1] if t1 > t0 else audio)
        return np.stack(outs)

    def decode(self, hidden):                   # SopranoDecoder.callAsFunction :263-283; hidden [B, L, C]
        h = np.transpose(np.asarray(hidden, F), (0, 2, 1))
        L = h.shape[2]
        h = interpolate1d(h, self.cfg.upscale * (L - 1) + 1)
        return self.head(self.backbone(np.transpose(h, (0, 2, 1))))


def soprano_repetition_penalty(logits: np.ndarray, tokens, penalty: float) -> np.ndarray:
    """applyRepetitionPenalty, Soprano.swift:888-901: float32, once PER OCCURRENCE, logit > 0 ? /pen : *pen."""
    out = np.array(logits, F, copy=True)
    for t in tokens:
        if t < out.shape[0]:
            out[t] = out[t] / F(penalty) if out[t] > 0 else out[t] * F(penalty)
    return out


def make_synthetic_weights(cfg: SopranoDecoderConfig, seed: int = 99) -> dict:
    from . import synth
    W, key = {}, [seed * 100000]

    def t(shape, amp):
        key[0] += 1
        return synth.synth_tensor(key[0], shape, amp)

    p, d, inter = "decoder.decoder", cfg.decoder_dim, cfg.decoder_intermediate_dim
    W[p + ".embed.weight"] = t((d, cfg.input_kernel, cfg.hidden_size), math.sqrt(3.0 / (cfg.input_kernel * cfg.hidden_size)))
    W[p + ".embed.bias"] = t((d,), 0.05)
    W[p + ".norm.weight"] = (1.0 + t((d,), 0.1)).astype(F); W[p + ".norm.bias"] = t((d,), 0.05)
    for i in range(cfg.decoder_num_layers):
        q = f"{p}.convnext.{i}"
        W[q + ".dwconv.weight"] = t((d, cfg.dw_kernel, 1), math.sqrt(3.0 / cfg.dw_kernel)); W[q + ".dwconv.bias"] = t((d,), 0.05)
        W[q + ".norm.weight"] = (1.0 + t((d,), 0.1)).astype(F); W[q + ".norm.bias"] = t((d,), 0.05)
        W[q + ".pwconv1.weight"] = t((inter, d), math.sqrt(3.0 / d)); W[q + ".pwconv1.bias"] = t((inter,), 0.05)
        W[q + ".pwconv2.weight"] = t((d, inter), math.sqrt(3.0 / inter)); W[q + ".pwconv2.bias"] = t((d,), 0.05)
        W[q + ".gamma"] = (0.5 + t((d,), 0.3)).astype(F)
    W[p + ".final_layer_norm.weight"] = (1.0 + t((d,), 0.1)).astype(F); W[p + ".final_layer_norm.bias"] = t((d,), 0.05)
    W["decoder.head.out.weight"] = t((cfg.n_fft + 2, d), 0.6 * math.sqrt(3.0 / d))
    W["decoder.head.out.bias"] = t((cfg.n_fft + 2,), 0.05)
    return W
