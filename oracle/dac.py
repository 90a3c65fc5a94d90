"""Descript DAC codec (codes -> waveform, and waveform -> codes): CPU restatement of the reference.  Test infrastructure only.

Follows Sources/MLXAudioCodecs/Descript/DescriptDAC.swift:7-160,172-245 (DescriptResidualUnit, DescriptDecoderBlock,
DescriptDecoder, DescriptDAC.decodeFromCodes), DescriptQuantization.swift:13-24,74-81,150-163 (Snake, decodeCode, fromCodes) and
BigVGAN/BigVGANLayers.swift:6-9,113-225 (weight-normalised conv / transposed conv: w = g * v / (||v|| + 1e-12), the norm taken
over every axis except 0 for Conv1d and except 2 (input channels) for ConvTranspose1d; output_padding = 1 IS forwarded to
convTransposed1d here, unlike SNAC).  Layout in this file is [B, C, T]; weights keep the MLX shapes ([out, k, in]).
[3P] MLX.convTransposed1d: full result y[n*s + j] += x[n, c] w[o, j, c], then `padding` samples dropped on the left and
`padding - output_padding` on the right.  Pinned by the reference's own shape tests (Tests/MLXAudioCodecsTests.swift:1127-1194:
250 frames -> 80 043 samples at rates [8,5,4,2]; 430 -> 220 235 at [8,8,4,2])."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as TF

F = np.float32


@dataclass
class DacConfig:                         # DescriptDACConfig.swift:3-34
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


TINY = DacConfig(encoder_dim=4, encoder_rates=(2, 2), latent_dim=24, decoder_dim=48, decoder_rates=(3, 2, 4), n_codebooks=3,
                 codebook_size=40, codebook_dim=8)


def convt_out_len(T: int, s: int) -> int:
    k, pad = 2 * s, math.ceil(s / 2)
    return (T - 1) * s - 2 * pad + (k - 1) + 1 + 1          # output_padding = 1


def num_samples(cfg: DacConfig, T: int) -> int:
    for s in cfg.decoder_rates:
        T = convt_out_len(T, s)
    return T


def _wn(g, v, except_dim):
    axes = tuple(a for a in range(v.ndim) if a != except_dim)
    return g * v / (torch.sqrt(torch.sum(v * v, dim=axes, keepdim=True)) + 1e-12)


def _snake(x, alpha):
    a = alpha.reshape(1, -1, 1)
    s = torch.sin(a * x)
    return x + (1.0 / (a + 1e-9)) * (s * s)


class DacOracle:
    def __init__(self, cfg: DacConfig, weights: dict):
        self.cfg = cfg
        self.w = {k: torch.as_tensor(np.asarray(v, F)) for k, v in weights.items()}

    def conv(self, p, x, pad, dil=1):
        w = _wn(self.w[p + ".weight_g"], self.w[p + ".weight_v"], 0)
        return TF.conv1d(x, w.permute(0, 2, 1).contiguous(), self.w[p + ".bias"], padding=pad, dilation=dil)

    def convt(self, p, x, s):
        w = _wn(self.w[p + ".weight_g"], self.w[p + ".weight_v"], 2)       # [out, k, in]
        return TF.conv_transpose1d(x, w.permute(2, 0, 1).contiguous(), self.w[p + ".bias"], stride=s, padding=math.ceil(s / 2),
                                   output_padding=1)

    def from_codes(self, codes):
        """codes [B, n_cb, T] -> z_q [B, latent, T] (DescriptQuantization.swift:150-163)."""
        codes = torch.as_tensor(np.asarray(codes, np.int64))
        z = 0
        for i in range(codes.shape[1]):
            p = f"quantizer.quantizers.{i}"
            e = self.w[p + ".codebook.weight"][codes[:, i]].transpose(1, 2)      # [B, 8, T]
            z = z + self.conv(p + ".outProj", e, 0)
        return z

    def decode(self, z, stop_after=None):
        cfg = self.cfg
        with torch.no_grad():
            h = self.conv("decoder.model.0", z, 3)
            for bi, s in enumerate(cfg.decoder_rates):
                p = f"decoder.model.{bi + 1}.block"
                h = self.convt(p + ".1", _snake(h, self.w[p + ".0.alpha"]), s)
                for ri, dil in enumerate((1, 3, 9)):
                    q = f"{p}.{ri + 2}.block"
                    t = self.conv(q + ".1", _snake(h, self.w[q + ".0.alpha"]), 3 * dil, dil)
                    h = h + self.conv(q + ".3", _snake(t, self.w[q + ".2.alpha"]), 0)
                if stop_after == f"block{bi}":
                    return h.numpy()
            n = len(cfg.decoder_rates)
            h = self.conv(f"decoder.model.{n + 2}", _snake(h, self.w[f"decoder.model.{n + 1}.alpha"]), 3)
            return torch.tanh(h)[:, 0].numpy()

    def decode_from_codes(self, codes, stop_after=None):
        with torch.no_grad():
            return self.decode(self.from_codes(codes), stop_after)

    # ---- encode (DescriptDAC.swift:40-95 DescriptEncoderBlock / DescriptEncoder, :216-233 preprocess + encode;
    # DescriptQuantization.swift:54-94 VectorQuantize: in_proj -> nearest L2-normalised code -> out_proj, :121-147 residual loop)
    def preprocess(self, audio):
        """right-pad to a multiple of the hop length (prod(encoder_rates)); audio [B, n] -> [B, 1, n']"""
        a = torch.as_tensor(np.asarray(audio, F))
        hop = int(np.prod(self.cfg.encoder_rates))
        n = a.shape[1]
        return TF.pad(a, (0, (-n) % hop))[:, None, :]

    def encoder(self, x):
        cfg = self.cfg
        h = self.conv("encoder.block.0", x, 3)
        for bi, s in enumerate(cfg.encoder_rates):
            p = f"encoder.block.{bi + 1}.block"
            for ri, dil in enumerate((1, 3, 9)):
                q = f"{p}.{ri}.block"
                t = self.conv(q + ".1", _snake(h, self.w[q + ".0.alpha"]), 3 * dil, dil)
                h = h + self.conv(q + ".3", _snake(t, self.w[q + ".2.alpha"]), 0)
            w = _wn(self.w[p + ".4.weight_g"], self.w[p + ".4.weight_v"], 0)
            h = TF.conv1d(_snake(h, self.w[p + ".3.alpha"]), w.permute(0, 2, 1).contiguous(), self.w[p + ".4.bias"], stride=s,
                          padding=math.ceil(s / 2))
        n = len(cfg.encoder_rates)
        return self.conv(f"encoder.block.{n + 2}", _snake(h, self.w[f"encoder.block.{n + 1}.alpha"]), 1)

    def encode(self, audio, n_quantizers=None, return_latent=False, return_margins=False):
        """audio [B, n] -> codes [B, n_cb, T] int64 (and z [B, latent, T] before quantisation; margins = per (b, q, t) gap between the
        best and the second-best code's distance: a device may legitimately differ where it is at float32 rounding level)."""
        with torch.no_grad():
            z = self.encoder(self.preprocess(audio))
            resid, codes, margins = z, [], []
            for i in range(n_quantizers or self.cfg.n_codebooks):
                p = f"quantizer.quantizers.{i}"
                ze = self.conv(p + ".inProj", resid, 0)                                   # [B, cd, T]
                B, cd, T = ze.shape
                enc = ze.transpose(1, 2).reshape(B * T, cd)
                cb = self.w[p + ".codebook.weight"]
                en = enc / torch.clamp(torch.sqrt((enc * enc).sum(1, keepdim=True)), min=1e-12)
                cn = cb / torch.clamp(torch.sqrt((cb * cb).sum(1, keepdim=True)), min=1e-12)
                dist = (en * en).sum(1, keepdim=True) - 2 * en @ cn.t() + (cn * cn).sum(1, keepdim=True).t()
                idx = torch.argmax(-dist, dim=1)
                top2 = torch.topk(-dist, 2, dim=1).values
                margins.append((top2[:, 0] - top2[:, 1]).reshape(B, T))
                idx = idx.reshape(B, T)
                codes.append(idx)
                zq = self.conv(p + ".outProj", cb[idx].transpose(1, 2), 0)
                resid = resid - zq
            out = torch.stack(codes, 1).numpy()
            res = [out]
            if return_latent:
                res.append(z.numpy())
            if return_margins:
                res.append(torch.stack(margins, 1).numpy())
            return res[0] if len(res) == 1 else tuple(res)


def make_synthetic_weights(cfg: DacConfig, seed: int = 808) -> dict:
    from . import synth
    W, key = {}, [seed * 100000]

    def t(shape, amp, offset=0.0):
        key[0] += 1
        return (synth.synth_tensor(key[0], shape, amp) + F(offset)).astype(F)

    def wn(p, co, k, ci, transposed=False, gain=1.0):
        v = t((co, k, ci), gain * math.sqrt(3.0 / (k * ci)))
        if transposed:
            nrm = np.sqrt((v * v).sum(axis=(0, 1), keepdims=True))
            g = nrm * t((1, 1, ci), 0.3, 1.0)
        else:
            nrm = np.sqrt((v * v).sum(axis=(1, 2), keepdims=True))
            g = nrm * t((co, 1, 1), 0.3, 1.0)
        W[p + ".weight_v"], W[p + ".weight_g"], W[p + ".bias"] = v, g.astype(F), t((co,), 0.05)

    D = cfg.resolved_latent
    for i in range(cfg.n_codebooks):
        p = f"quantizer.quantizers.{i}"
        W[p + ".codebook.weight"] = t((cfg.codebook_size, cfg.codebook_dim), math.sqrt(3.0) / math.sqrt(cfg.n_codebooks))
        wn(p + ".outProj", D, 1, cfg.codebook_dim)
    wn("decoder.model.0", cfg.decoder_dim, 7, D)
    for bi, s in enumerate(cfg.decoder_rates):
        cin, cout = cfg.decoder_dim >> bi, cfg.decoder_dim >> (bi + 1)
        p = f"decoder.model.{bi + 1}.block"
        W[p + ".0.alpha"] = t((1, 1, cin), 0.75, 1.25)
        wn(p + ".1", cout, 2 * s, cin, transposed=True, gain=math.sqrt(s) * 0.8)
        for ri in range(3):
            q = f"{p}.{ri + 2}.block"
            W[q + ".0.alpha"] = t((1, 1, cout), 0.75, 1.25)
            wn(q + ".1", cout, 7, cout, gain=0.7)
            W[q + ".2.alpha"] = t((1, 1, cout), 0.75, 1.25)
            wn(q + ".3", cout, 1, cout, gain=0.3)
    n, cl = len(cfg.decoder_rates), cfg.decoder_dim >> len(cfg.decoder_rates)
    W[f"decoder.model.{n + 1}.alpha"] = t((1, 1, cl), 0.75, 1.25)
    wn(f"decoder.model.{n + 2}", 1, 7, cl, gain=0.5)
    # encoder + in_proj (keys after DescriptDAC.sanitize)
    ch = cfg.encoder_dim
    wn("encoder.block.0", ch, 7, 1, gain=1.5)
    for bi, s in enumerate(cfg.encoder_rates):
        p = f"encoder.block.{bi + 1}.block"
        for ri in range(3):
            q = f"{p}.{ri}.block"
            W[q + ".0.alpha"] = t((1, 1, ch), 0.75, 1.25)
            wn(q + ".1", ch, 7, ch, gain=0.7)
            W[q + ".2.alpha"] = t((1, 1, ch), 0.75, 1.25)
            wn(q + ".3", ch, 1, ch, gain=0.3)
        W[p + ".3.alpha"] = t((1, 1, ch), 0.75, 1.25)
        wn(p + ".4", 2 * ch, 2 * s, ch, gain=1.2)
        ch *= 2
    m = len(cfg.encoder_rates)
    W[f"encoder.block.{m + 1}.alpha"] = t((1, 1, ch), 0.75, 1.25)
    wn(f"encoder.block.{m + 2}", D, 3, ch, gain=1.5)
    for i in range(cfg.n_codebooks):
        wn(f"quantizer.quantizers.{i}.inProj", cfg.codebook_dim, 1, D, gain=1.5)
    return W
