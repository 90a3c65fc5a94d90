"""Qwen3-TTS speaker encoder (ECAPA-TDNN x-vector of the reference audio): CPU restatement of the reference.  Test infrastructure
only: the device side is csrc/q3_reference.hip, held to this module by tests/test_gpu_q3_reference.py.

Follows Sources/MLXAudioTTS/Models/Qwen3TTS/Qwen3TTSSpeakerEncoder.swift: reflectPad1D (:6-16), TimeDelayNetBlock (:20-42: reflect pad
(k-1) d / 2 on both sides, Conv1d with dilation, ReLU), Res2NetBlock (:46-96: channel chunks, chunk i > 1 gets the previous chunk's
output added before its TDNN), SqueezeExcitationBlock (:100-129: mean over time -> 1x1 -> ReLU -> 1x1 -> sigmoid gate),
SqueezeExcitationRes2NetBlock (:133-184: tdnn1 -> res2net -> tdnn2 -> SE, + residual), AttentiveStatisticsPooling (:188-233: global
mean / std appended to every frame, TDNN -> tanh -> 1x1 -> softmax over time, attention-weighted mean and std), the encoder (:237-307:
first TDNN, SE-Res2Net blocks, concatenation of the blocks' outputs (not the first TDNN's), mfa TDNN, pooling, fc).  Input = the log-mel
of computeMelSpectrogram(sampleRate 24000, nFft 1024, hop 256, 128 mels) (Qwen3TTS.swift:839-880), [B, T, 128]; output [B, enc_dim].
Layout here is [B, C, T]; conv weights are stored in the MLX layout [out, k, in] (the sanitised checkpoint, :309-331)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

F = np.float32


@dataclass
class EcapaConfig:                                   # Qwen3TTSSpeakerEncoderConfig (Qwen3TTSConfig.swift:69-117)
    mel_dim: int = 128
    enc_dim: int = 1024
    enc_channels: tuple = (512, 512, 512, 512, 1536)
    enc_kernel_sizes: tuple = (5, 3, 3, 3, 1)
    enc_dilations: tuple = (1, 2, 3, 4, 1)
    enc_attention_channels: int = 128
    enc_res2net_scale: int = 8
    enc_se_channels: int = 128
    sample_rate: int = 24000


TINY = EcapaConfig(mel_dim=12, enc_dim=20, enc_channels=(16, 16, 16, 48), enc_kernel_sizes=(5, 3, 3, 1), enc_dilations=(1, 2, 3, 1),
                   enc_attention_channels=8, enc_res2net_scale=4, enc_se_channels=6)


def reflect_pad(x, pad):
    """reflectPad1D (:6-16) on [B, C, T]: pad clamped to T - 1, nothing for T <= 1."""
    T = x.shape[-1]
    if pad <= 0 or T <= 1:
        return x
    p = min(pad, T - 1)
    return np.concatenate([x[..., 1:p + 1][..., ::-1], x, x[..., -(p + 1):-1][..., ::-1]], axis=-1)


def conv1d(x, w, b, dilation=1):
    """MLX Conv1d without padding on [B, C, T]; w [out, k, in]."""
    B, C, T = x.shape
    O, K, _ = w.shape
    Tout = T - (K - 1) * dilation
    y = np.zeros((B, O, Tout), F)
    for j in range(K):
        y += np.einsum("oc,bct->bot", w[:, j, :], x[:, :, j * dilation: j * dilation + Tout]).astype(F)
    return y + b[None, :, None]


class EcapaOracle:
    def __init__(self, cfg: EcapaConfig, weights: dict):
        self.cfg = cfg
        self.w = {k: np.asarray(v, F) for k, v in weights.items()}

    def tdnn(self, p, x, k, d):
        y = conv1d(reflect_pad(x, (k - 1) * d // 2), self.w[p + ".conv.weight"], self.w[p + ".conv.bias"], d)
        return np.maximum(y, F(0))

    def res2net(self, p, x, k, d):
        scale = self.cfg.enc_res2net_scale
        chunks = np.split(x, scale, axis=1)
        outs, prev = [], None
        for i, ch in enumerate(chunks):
            if i == 0:
                prev = ch
            elif i == 1:
                prev = self.tdnn(f"{p}.blocks.{i - 1}", ch, k, d)
            else:
                prev = self.tdnn(f"{p}.blocks.{i - 1}", ch + prev, k, d)
            outs.append(prev)
        return np.concatenate(outs, axis=1)

    def se(self, p, x):
        s = x.mean(axis=2, keepdims=True, dtype=F)
        s = np.maximum(conv1d(s, self.w[p + ".conv1.weight"], self.w[p + ".conv1.bias"]), F(0))
        s = conv1d(s, self.w[p + ".conv2.weight"], self.w[p + ".conv2.bias"])
        return x * (F(1) / (F(1) + np.exp(-s)))

    def se_res2net_block(self, p, x, k, d):
        y = self.tdnn(p + ".tdnn1", x, 1, 1)
        y = self.res2net(p + ".res2net_block", y, k, d)
        y = self.tdnn(p + ".tdnn2", y, 1, 1)
        return self.se(p + ".se_block", y) + x

    def asp(self, p, x):
        eps = F(1e-12)
        T = x.shape[2]
        mu = x.mean(axis=2, keepdims=True, dtype=F)
        sd = np.sqrt(((x - mu) ** 2).mean(axis=2, keepdims=True, dtype=F) + eps)
        a = np.concatenate([x, np.broadcast_to(mu, x.shape), np.broadcast_to(sd, x.shape)], axis=1)
        a = np.tanh(self.tdnn(p + ".tdnn", a, 1, 1))
        a = conv1d(a, self.w[p + ".conv.weight"], self.w[p + ".conv.bias"])
        a = a - a.max(axis=2, keepdims=True)
        a = np.exp(a)
        a = a / a.sum(axis=2, keepdims=True, dtype=F)
        m = (a * x).sum(axis=2, keepdims=True, dtype=F)
        v = (a * (x - m) * (x - m)).sum(axis=2, keepdims=True, dtype=F)
        return np.concatenate([m, np.sqrt(np.maximum(v, eps))], axis=1)

    def __call__(self, mel, return_intermediates=False):
        """mel [B, T, mel_dim] -> x-vector [B, enc_dim]."""
        cfg = self.cfg
        x = np.transpose(np.asarray(mel, F), (0, 2, 1))
        hs, inter = [], {}
        n = len(cfg.enc_channels)
        x = self.tdnn("blocks.0", x, cfg.enc_kernel_sizes[0], cfg.enc_dilations[0])
        hs.append(x); inter["block0"] = x
        for i in range(1, n - 1):
            x = self.se_res2net_block(f"blocks.{i}", x, cfg.enc_kernel_sizes[i], cfg.enc_dilations[i])
            hs.append(x); inter[f"block{i}"] = x
        if len(hs) >= 2:
            x = np.concatenate(hs[1:], axis=1)
        x = self.tdnn("mfa", x, cfg.enc_kernel_sizes[-1], cfg.enc_dilations[-1])
        inter["mfa"] = x
        x = self.asp("asp", x)
        inter["asp"] = x
        x = conv1d(x, self.w["fc.weight"], self.w["fc.bias"])[:, :, 0]
        return (x, inter) if return_intermediates else x


def make_synthetic_weights(cfg: EcapaConfig, seed: int = 31) -> dict:
    from . import synth
    W, key = {}, [seed * 100000]

    def conv(p, co, k, ci, gain=1.0):
        key[0] += 2
        W[p + ".weight"] = synth.synth_tensor(key[0], (co, k, ci), gain * np.sqrt(3.0 / (k * ci)))
        W[p + ".bias"] = synth.synth_tensor(key[0] + 1, (co,), 0.1)
    ch, ks = cfg.enc_channels, cfg.enc_kernel_sizes
    conv("blocks.0.conv", ch[0], ks[0], cfg.mel_dim, 1.5)
    for i in range(1, len(ch) - 1):
        p = f"blocks.{i}"
        conv(p + ".tdnn1.conv", ch[i], 1, ch[i - 1], 1.5)
        for j in range(cfg.enc_res2net_scale - 1):
            conv(f"{p}.res2net_block.blocks.{j}.conv", ch[i] // cfg.enc_res2net_scale, ks[i], ch[i] // cfg.enc_res2net_scale, 1.5)
        conv(p + ".tdnn2.conv", ch[i], 1, ch[i], 1.5)
        conv(p + ".se_block.conv1", cfg.enc_se_channels, 1, ch[i])
        conv(p + ".se_block.conv2", ch[i], 1, cfg.enc_se_channels)
    cat = sum(ch[1:-1]) if len(ch) > 2 else ch[0]
    conv("mfa.conv", ch[-1], ks[-1], cat, 1.5)
    conv("asp.tdnn.conv", cfg.enc_attention_channels, 1, 3 * ch[-1])
    conv("asp.conv", ch[-1], 1, cfg.enc_attention_channels, 2.0)
    conv("fc", cfg.enc_dim, 1, 2 * ch[-1])
    return W
