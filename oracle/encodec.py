"""EnCodec decoder (codes -> waveform): CPU restatement of the reference.  Test infrastructure only.

Follows Sources/MLXAudioCodecs/Encodec/: EncodecDecoder (Encodec.swift:94-170), decodeFrame / linearOverlapAdd / decode
(:295-398), EncodecLSTM / EncodecLSTMBlock (EncodecLayers.swift:15-80), EncodecConv1d incl. its padding rules (:84-214:
padding_total = kernel - stride (dilation NOT included, as in HF transformers), causal = everything on the left, reflect
padding with the index rule min(pad - i, n - 1)), EncodecConvTranspose1dLayer (:218-262: full transposed conv, then
padding_total * trim_right_ratio samples trimmed on the right when causal), EncodecResnetBlock (:266-325, conv shortcut),
ELU (:340-350), EncodecResidualVectorQuantizer.decode (EncodecQuantization.swift:117-133: sum of codebook rows).
The reference executes its transposed conv as scalar Swift loops over asArray copies (EncodecLayers.swift:395-420) and the LSTM
as a Python-style per-step loop; arithmetic is plain float32.  norm_type "weight_norm" (the 24 kHz model: plain convs) and
"time_group_norm" (the 48 kHz stereo model: GroupNorm(1 group, pytorchCompatible) after EVERY conv and - in the transposed-conv
layer - BEFORE the padding is trimmed, :128-131,203-212,244-262; non-causal padding split) are restated.  Layout here is [B, C, T]."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as TF

F = np.float32


@dataclass
class EncodecConfig:                      # EncodecConfig.swift:64-89 (encodec_24khz defaults)
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
    use_conv_shortcut: bool = True

    @property
    def hop_length(self) -> int:
        return int(np.prod(self.upsampling_ratios))

    @property
    def num_quantizers(self) -> int:      # EncodecQuantization.swift:60-64
        frame_rate = int(math.ceil(self.sampling_rate / self.hop_length))
        return int(1000 * max(self.target_bandwidths) / (frame_rate * 10))


# the 48 kHz model in miniature: stereo, non-causal padding, GroupNorm after every conv
TINY_48K = EncodecConfig(audio_channels=2, num_filters=4, codebook_size=32, codebook_dim=16, hidden_size=16, upsampling_ratios=(3, 2, 2),
                         target_bandwidths=(1.5, 3.0), sampling_rate=1200, use_causal_conv=False, norm_type="time_group_norm")

TINY = EncodecConfig(num_filters=4, codebook_size=32, codebook_dim=16, hidden_size=16, upsampling_ratios=(3, 2, 2),
                     target_bandwidths=(1.5, 3.0), sampling_rate=1200)


def pad1d(x, left, right, mode):
    """EncodecConv1d.pad1d (:130-171) on [B, C, T]."""
    if mode != "reflect":
        return TF.pad(x, (left, right))
    n = x.shape[-1]
    parts = []
    if left > 0:
        parts.append(x[..., [min(left - i, n - 1) for i in range(left)]])
    parts.append(x)
    if right > 0:
        parts.append(x[..., [max(n - 2 - i, 0) for i in range(right)]])
    return torch.cat(parts, -1)


class EncodecOracle:
    def __init__(self, cfg: EncodecConfig, weights: dict):
        self.cfg = cfg
        self.w = {k: torch.as_tensor(np.asarray(v, F)) for k, v in weights.items()}

    def conv(self, p, x, k, stride=1, dilation=1):
        cfg = self.cfg
        keff, ptotal = (k - 1) * dilation + 1, k - stride
        n = x.shape[-1]
        nframes = (n - keff + ptotal) / stride + 1
        ideal = (int(math.ceil(nframes)) - 1) * stride + keff - ptotal
        extra = max(0, ideal - n)
        if cfg.use_causal_conv:
            xp = pad1d(x, ptotal, extra, cfg.pad_mode)
        else:
            r = ptotal // 2
            xp = pad1d(x, ptotal - r, r + extra, cfg.pad_mode)
        return self.norm(p, TF.conv1d(xp, self.w[p + ".conv.weight"].permute(0, 2, 1).contiguous(), self.w[p + ".conv.bias"],
                                      stride=stride, dilation=dilation))

    def norm(self, p, h):
        """GroupNorm(groupCount: 1, dimensions: C, pytorchCompatible: true) (EncodecLayers.swift:128-131): statistics over (C, T)."""
        if self.cfg.norm_type != "time_group_norm":
            return h
        mu = h.mean(dim=(1, 2), keepdim=True)
        var = ((h - mu) ** 2).mean(dim=(1, 2), keepdim=True)
        return (h - mu) / torch.sqrt(var + 1e-5) * self.w[p + ".norm.weight"][None, :, None] + self.w[p + ".norm.bias"][None, :, None]

    def conv_transpose(self, p, x, k, stride):
        cfg = self.cfg
        y = TF.conv_transpose1d(x, self.w[p + ".conv.weight"].permute(2, 0, 1).contiguous(), self.w[p + ".conv.bias"], stride=stride)
        y = self.norm(p, y)                                     # BEFORE the trim (:244-262)
        ptotal = k - stride
        right = int(math.ceil(ptotal * cfg.trim_right_ratio)) if cfg.use_causal_conv else ptotal // 2
        left = ptotal - right
        end = y.shape[-1] - right
        return y[..., left:end] if end > left else y

    def lstm(self, p, x):
        """EncodecLSTM (:15-62), x [B, T, C]; gates i, f, g, o."""
        Wx, Wh, b = self.w[p + ".Wx"], self.w[p + ".Wh"], self.w[p + ".bias"]
        H = Wh.shape[1]
        xp = x @ Wx.t() + b
        B, T, _ = x.shape
        h, c, out = None, torch.zeros(B, H), []
        for t in range(T):
            g = xp[:, t] + (h @ Wh.t() if h is not None else 0.0)
            i, f, gg, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H]), torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:])
            c = f * c + i * gg
            h = o * torch.tanh(c)
            out.append(h)
        return torch.stack(out, 1)

    def quantizer_decode(self, codes):
        codes = torch.as_tensor(np.asarray(codes, np.int64))
        q = 0
        for i in range(codes.shape[1]):
            q = q + self.w[f"quantizer.layers.{i}.codebook.embed"][codes[:, i]]      # [B, T, D]
        return q.transpose(1, 2)

    def decoder(self, z, stop_after=None):
        cfg = self.cfg
        h = self.conv("decoder.layers.0", z, cfg.kernel_size)
        if stop_after == "conv0":
            return h
        y = h.transpose(1, 2)
        r = y
        for j in range(cfg.num_lstm_layers):
            r = self.lstm(f"decoder.layers.1.lstm.{j}", r)
        h = (r + y).transpose(1, 2)
        if stop_after == "lstm":
            return h
        li = 2
        for bi, ratio in enumerate(cfg.upsampling_ratios):
            h = self.conv_transpose(f"decoder.layers.{li + 1}", TF.elu(h), 2 * ratio, ratio)
            li += 2
            for j in range(cfg.num_residual_layers):
                p = f"decoder.layers.{li}"
                dil = cfg.dilation_growth_rate ** j
                t = self.conv(p + ".block.1", TF.elu(h), cfg.residual_kernel_size, dilation=dil)
                t = self.conv(p + ".block.3", TF.elu(t), 1)
                sc = self.conv(p + ".shortcut", h, 1) if cfg.use_conv_shortcut else h
                h = sc + t
                li += 1
            if stop_after == f"block{bi}":
                return h
        return self.conv(f"decoder.layers.{li + 1}", TF.elu(h), cfg.last_kernel_size)

    def decode_frame(self, codes, scale=None, stop_after=None):
        """decodeFrame (:295-302): codes [B, nq, T] -> [B, T * hop] (audio_channels = 1) or [B, channels, T * hop]."""
        with torch.no_grad():
            out = self.decoder(self.quantizer_decode(codes), stop_after)
            if stop_after is not None:
                return out.numpy()
            if scale is not None:
                out = out * torch.as_tensor(np.asarray(scale, F)).reshape(-1, 1, 1)
            return out[:, 0].numpy() if self.cfg.audio_channels == 1 else out.numpy()


def linear_overlap_add(frames, hop_stride):
    """Encodec.linearOverlapAdd (:304-355): frames list of [B, L_i]; triangular weights of the FIRST frame's length."""
    L = frames[0].shape[1]
    total = hop_stride * (len(frames) - 1) + frames[-1].shape[1]
    tv = (np.arange(L, dtype=F) + F(1)) / F(L + 1)
    wv = (F(0.5) - np.abs(tv - F(0.5))).astype(F)
    out = np.zeros((frames[0].shape[0], total), F)
    sw = np.zeros(total, F)
    off = 0
    for f in frames:
        n = f.shape[1]
        out[:, off:off + n] += wv[:n] * f
        sw[off:off + n] += wv[:n]
        off += hop_stride
    nz = sw != 0
    out[:, nz] /= sw[nz]
    return out


def make_synthetic_weights(cfg: EncodecConfig, seed: int = 909, n_quantizers: int | None = None) -> dict:
    from . import synth
    W, key = {}, [seed * 100000]

    def t(shape, amp):
        key[0] += 1
        return synth.synth_tensor(key[0], shape, amp)

    def conv(p, co, k, ci, gain=1.0):
        W[p + ".conv.weight"] = t((co, k, ci), gain * math.sqrt(3.0 / (k * ci)))
        W[p + ".conv.bias"] = t((co,), 0.05)
        if cfg.norm_type == "time_group_norm":
            W[p + ".norm.weight"] = (1.0 + t((co,), 0.3)).astype(np.float32)
            W[p + ".norm.bias"] = t((co,), 0.1)

    nq = n_quantizers or cfg.num_quantizers
    for i in range(nq):
        W[f"quantizer.layers.{i}.codebook.embed"] = t((cfg.codebook_size, cfg.codebook_dim), math.sqrt(3.0) / math.sqrt(nq))
    scaling = 2 ** len(cfg.upsampling_ratios)
    dim = scaling * cfg.num_filters
    conv("decoder.layers.0", dim, cfg.kernel_size, cfg.hidden_size)
    for j in range(cfg.num_lstm_layers):
        p = f"decoder.layers.1.lstm.{j}"
        W[p + ".Wx"] = t((4 * dim, dim), math.sqrt(3.0 / dim))
        W[p + ".Wh"] = t((4 * dim, dim), math.sqrt(3.0 / dim))
        W[p + ".bias"] = t((4 * dim,), 0.1)
    li = 2
    for ratio in cfg.upsampling_ratios:
        conv(f"decoder.layers.{li + 1}", dim // 2, 2 * ratio, dim, gain=math.sqrt(ratio))
        li += 2
        dim //= 2
        for j in range(cfg.num_residual_layers):
            p = f"decoder.layers.{li}"
            conv(p + ".block.1", dim // cfg.compress, cfg.residual_kernel_size, dim)
            conv(p + ".block.3", dim, 1, dim // cfg.compress, gain=0.5)
            if cfg.use_conv_shortcut:
                conv(p + ".shortcut", dim, 1, dim)
            li += 1
    conv(f"decoder.layers.{li + 1}", cfg.audio_channels, cfg.last_kernel_size, dim, gain=0.5)
    return W
