"""Qwen3-TTS speech-tokenizer ENCODER (reference audio -> codec codes, the in-context voice-cloning prompt): CPU restatement of the
reference.  Test infrastructure only: the device side is csrc/q3_reference.hip, held to this module by tests/test_gpu_q3_reference.py.

Follows Qwen3TTSSpeechTokenizerEncoder (Qwen3TTSSpeechTokenizer.swift:792-881), which is the Mimi encoder of
Sources/MLXAudioCodecs/Mimi/: StreamableConv1d (Conv.swift:171-227: padding_total = (k-1) d + 1 - stride, everything on the left when
causal, plus the extra right padding that completes the last stride; constant = zeros, ConvDownsample1d uses edge replication),
SeanetResnetBlock / EncoderLayer / SeanetEncoder (Seanet.swift:92-258: ELU before every conv, residual add with true skip, the
ratios applied in REVERSED order, channels doubling), Attention / MlpNoGating / TransformerLayer / ProjectedTransformer
(Transformer.swift:107-369: fused in_proj without bias, traditional = interleaved RoPE, plain causal mask for a whole-sequence call -
the `context` window only trims the streaming cache -, LayerNorm eps 1e-5, exact GELU, per-channel layer scale), ConvDownsample1d
(Conv.swift:346-359: k = 2 stride, no bias), EuclideanCodebook / VectorQuantization / ResidualVectorQuantization /
SplitResidualVectorQuantizer.encode (Quantization.swift:6-211: embedding = embedding_sum / max(cluster_usage, 1e-5), nearest code =
argmin(|e|^2 / 2 - x.e), first index on ties; one semantic quantizer and nq - 1 acoustic ones, each group behind its own 1x1 input
projection, residuals inside a group).  Layout [B, C, T]; conv weights in the MLX layout [out, k, in]; Linear [out, in]."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as TF

F = np.float32


@dataclass
class MimiEncoderConfig:                          # Qwen3TTSTokenizerEncoderConfig (Qwen3TTSConfig.swift), Mimi defaults
    audio_channels: int = 1
    num_filters: int = 64
    kernel_size: int = 7
    last_kernel_size: int = 3
    residual_kernel_size: int = 3
    num_residual_layers: int = 1
    dilation_growth_rate: int = 2
    compress: int = 2
    upsampling_ratios: tuple = (8, 6, 5, 4)
    use_causal_conv: bool = True
    use_conv_shortcut: bool = False
    hidden_size: int = 512
    num_hidden_layers: int = 8
    num_attention_heads: int = 8
    intermediate_size: int = 2048
    layer_scale_initial_scale: float = 0.01
    rope_theta: float = 10000.0
    sliding_window: int = 250
    sampling_rate: int = 24000
    frame_rate: float = 12.5
    codebook_dim: int = 256
    codebook_size: int = 2048
    num_quantizers: int = 32
    valid_num_quantizers: int = 16              # Qwen3-TTS keeps the first 16 code groups (:876-880)

    @property
    def downsample_stride(self) -> int:           # :805-807
        enc_rate = self.sampling_rate / float(np.prod(self.upsampling_ratios))
        return max(1, int(enc_rate / self.frame_rate))


TINY = MimiEncoderConfig(num_filters=4, upsampling_ratios=(3, 2), hidden_size=16, num_hidden_layers=2, num_attention_heads=2,
                         intermediate_size=32, sampling_rate=240, frame_rate=20.0, codebook_dim=8, codebook_size=32, num_quantizers=5,
                         valid_num_quantizers=4)


class MimiEncoderOracle:
    def __init__(self, cfg: MimiEncoderConfig, weights: dict):
        self.cfg = cfg
        self.w = {k: torch.as_tensor(np.asarray(v, F)) for k, v in weights.items()}

    # -- StreamableConv1d.callAsFunction (Conv.swift:206-226)
    def sconv(self, p, x, k, stride=1, dilation=1, causal=True, mode="constant", bias=True):
        keff = (k - 1) * dilation + 1
        ptotal = keff - stride
        n = x.shape[-1]
        nframes = max(n + ptotal - keff, 0)
        ideal = (int(math.ceil(nframes / stride + 1.0)) - 1) * stride + keff - ptotal
        extra = max(0, ideal - n)
        left, right = (ptotal, 0) if causal else (ptotal - ptotal // 2, ptotal // 2)
        xp = TF.pad(x, (left, right + extra), mode="replicate" if mode == "edge" else "constant")
        w = self.w[p + ".conv.conv.weight"].permute(0, 2, 1).contiguous()
        return TF.conv1d(xp, w, self.w[p + ".conv.conv.bias"] if bias else None, stride=stride, dilation=dilation)

    # -- SeanetEncoder (Seanet.swift:203-258)
    def seanet(self, audio):
        cfg = self.cfg
        x = self.sconv("encoder.init_conv1d", audio, cfg.kernel_size, causal=cfg.use_causal_conv)
        for li, ratio in enumerate(reversed(cfg.upsampling_ratios)):
            p = f"encoder.layers.{li}"
            dil = 1
            for ri in range(cfg.num_residual_layers):
                q = f"{p}.residuals.{ri}"
                h = self.sconv(q + ".block.0", TF.elu(x), cfg.residual_kernel_size, dilation=dil, causal=cfg.use_causal_conv)
                h = self.sconv(q + ".block.1", TF.elu(h), 1, causal=cfg.use_causal_conv)
                sc = self.sconv(q + ".shortcut", x, 1, causal=cfg.use_causal_conv) if cfg.use_conv_shortcut else x
                x = h + sc
                dil *= cfg.dilation_growth_rate
            x = self.sconv(p + ".downsample", TF.elu(x), 2 * ratio, stride=ratio, causal=True)
        return self.sconv("encoder.final_conv1d", TF.elu(x), cfg.last_kernel_size, causal=cfg.use_causal_conv)

    # -- ProjectedTransformer on [B, C, T] (Transformer.swift:121-369); whole sequence from an empty cache
    def transformer(self, x):
        cfg = self.cfg
        h = x.transpose(1, 2)
        B, T, D = h.shape
        H, hd = cfg.num_attention_heads, D // cfg.num_attention_heads
        pos = torch.arange(T, dtype=torch.float32)
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        ang = pos[:, None] * inv[None, :]
        cos, sin = torch.cos(ang), torch.sin(ang)

        def rope(a):                                  # traditional (interleaved pairs), MLXFast.RoPE(traditional: true)
            a1, a2 = a[..., 0::2], a[..., 1::2]
            return torch.stack([a1 * cos - a2 * sin, a1 * sin + a2 * cos], dim=-1).reshape(a.shape)
        mask = torch.full((T, T), float("-inf")).triu(1)
        for li in range(cfg.num_hidden_layers):
            p = f"encoder_transformer.transformer.layers.{li}"
            n1 = TF.layer_norm(h, (D,), self.w[p + ".norm1.weight"], self.w[p + ".norm1.bias"], 1e-5)
            qkv = (n1 @ self.w[p + ".self_attn.in_proj.weight"].T).reshape(B, T, 3, H, hd)
            q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
            sc = (rope(q) @ rope(k).transpose(-1, -2)) / math.sqrt(hd) + mask
            o = (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(B, T, D)
            h = h + (o @ self.w[p + ".self_attn.out_proj.weight"].T) * self.w[p + ".layer_scale_1.scale"]
            n2 = TF.layer_norm(h, (D,), self.w[p + ".norm2.weight"], self.w[p + ".norm2.bias"], 1e-5)
            m = TF.gelu(n2 @ self.w[p + ".gating.linear1.weight"].T) @ self.w[p + ".gating.linear2.weight"].T
            h = h + m * self.w[p + ".layer_scale_2.scale"]
        return h.transpose(1, 2)

    # -- SplitResidualVectorQuantizer.encode (Quantization.swift:121-199)
    def _rvq_encode(self, p, x, nq):
        x = TF.conv1d(x, self.w[p + ".input_proj.weight"].permute(0, 2, 1).contiguous())
        codes, resid = [], x.transpose(1, 2)                                          # [B, T, d]
        for i in range(nq):
            q = f"{p}.vq.layers.{i}.codebook"
            emb = self.w[q + ".embedding_sum"] / torch.clamp(self.w[q + ".cluster_usage"], min=1e-5)[:, None]
            dist = (emb * emb).sum(-1) / 2 - resid @ emb.T
            idx = torch.argmin(dist, dim=-1)
            resid = resid - emb[idx]
            codes.append(idx)
        return torch.stack(codes, dim=1)                                              # [B, nq, T]

    def encode(self, audio, return_hidden=False):
        """audio [B, 1, n] -> codes int64 [B, valid_num_quantizers, T]  (Qwen3TTSSpeechTokenizerEncoder.encode :868-880)."""
        cfg = self.cfg
        with torch.no_grad():
            x = self.seanet(torch.as_tensor(np.asarray(audio, F)))
            x = self.transformer(x)
            x = self.sconv("downsample.conv", x, 2 * cfg.downsample_stride, stride=cfg.downsample_stride, causal=cfg.use_causal_conv, mode="edge",
                           bias=False)
            codes = self._rvq_encode("quantizer.rvq_first", x, 1)
            if cfg.num_quantizers > 1:
                codes = torch.cat([codes, self._rvq_encode("quantizer.rvq_rest", x, cfg.num_quantizers - 1)], dim=1)
            codes = codes[:, : min(cfg.valid_num_quantizers, codes.shape[1])]
        return (codes.numpy(), x.numpy()) if return_hidden else codes.numpy()


def make_synthetic_weights(cfg: MimiEncoderConfig, seed: int = 77) -> dict:
    from . import synth
    W, key = {}, [seed * 100000]

    def t(shape, amp):
        key[0] += 1
        return synth.synth_tensor(key[0], shape, amp)

    def conv(p, co, k, ci, bias=True, gain=1.0):
        W[p + ".conv.conv.weight"] = t((co, k, ci), gain * math.sqrt(3.0 / (k * ci)))
        if bias:
            W[p + ".conv.conv.bias"] = t((co,), 0.05)
    nf, mult = cfg.num_filters, 1
    conv("encoder.init_conv1d", nf, cfg.kernel_size, cfg.audio_channels, gain=2.0)
    for li, ratio in enumerate(reversed(cfg.upsampling_ratios)):
        p = f"encoder.layers.{li}"
        dim = mult * nf
        for ri in range(cfg.num_residual_layers):
            conv(f"{p}.residuals.{ri}.block.0", dim // cfg.compress, cfg.residual_kernel_size, dim, gain=1.3)
            conv(f"{p}.residuals.{ri}.block.1", dim, 1, dim // cfg.compress, gain=0.7)
            if cfg.use_conv_shortcut:
                conv(f"{p}.residuals.{ri}.shortcut", dim, 1, dim)
        conv(p + ".downsample", 2 * dim, 2 * ratio, dim, gain=1.3)
        mult *= 2
    conv("encoder.final_conv1d", cfg.hidden_size, cfg.last_kernel_size, mult * nf, gain=1.3)
    D = cfg.hidden_size
    for li in range(cfg.num_hidden_layers):
        p = f"encoder_transformer.transformer.layers.{li}"
        for n in ("norm1", "norm2"):
            W[f"{p}.{n}.weight"] = (1.0 + t((D,), 0.2)).astype(F)
            W[f"{p}.{n}.bias"] = t((D,), 0.1)
        W[p + ".self_attn.in_proj.weight"] = t((3 * D, D), math.sqrt(3.0 / D))
        W[p + ".self_attn.out_proj.weight"] = t((D, D), math.sqrt(3.0 / D))
        W[p + ".gating.linear1.weight"] = t((cfg.intermediate_size, D), math.sqrt(3.0 / D))
        W[p + ".gating.linear2.weight"] = t((D, cfg.intermediate_size), math.sqrt(3.0 / cfg.intermediate_size))
        W[p + ".layer_scale_1.scale"] = (0.3 + t((D,), 0.1)).astype(F)
        W[p + ".layer_scale_2.scale"] = (0.3 + t((D,), 0.1)).astype(F)
    conv("downsample.conv", D, 2 * cfg.downsample_stride, D, bias=False)
    for grp, nq in (("rvq_first", 1), ("rvq_rest", cfg.num_quantizers - 1)):
        p = f"quantizer.{grp}"
        W[p + ".input_proj.weight"] = t((cfg.codebook_dim, 1, D), math.sqrt(3.0 / D))
        for i in range(nq):
            q = f"{p}.vq.layers.{i}.codebook"
            usage = (1.0 + np.abs(t((cfg.codebook_size,), 1.0))).astype(F)
            W[q + ".cluster_usage"] = usage
            W[q + ".embedding_sum"] = (t((cfg.codebook_size, cfg.codebook_dim), 1.0 / (i + 1)) * usage[:, None]).astype(F)
    return W
