"""SNAC codec decode: CPU restatement of the reference.  Test infrastructure only.

Follows (paths relative to Sources/MLXAudioCodecs/SNAC/):
  * ResidualVectorQuantize.fromCodes      VQ.swift:165-191   (+ decodeCode :88-94)
  * normalizeWeight / snake               Layers.swift:35-50
  * WNConv1d                              Layers.swift:54-118   (eps 1e-12 on the norm)
  * WNConvTranspose1d                     Layers.swift:122-183  (NO eps; output_padding ignored)
  * Snake1d / ResidualUnit / NoiseBlock   Layers.swift:188-232,263-279
  * DecoderBlock / Decoder                Layers.swift:283-315,364-421
  * SNAC.decode                           SNACDecoder.swift:127-131
Third-party semantics restated [3P, mlx-swift 0.31.4, not vendored]:
  MLX.conv1d         input NLC, weight [Cout, K, Cin/groups], cross-correlation:
                     y[t,co] = sum_k sum_ci x[t*stride + k*dil - pad, ci] * w[co,k,ci]
  MLX.convTransposed1d weight [Cout, K, Cin]:
                     y[t*stride + k*dil - pad, co] += x[t,ci] * w[co,k,ci]
                     length (T-1)*stride - 2*pad + dil*(K-1) + 1
Both are cross-checked against torch.nn.functional.conv1d / conv_transpose1d in
tests/test_oracle_snac.py (an independent implementation of the same definition).

NoiseBlock is stochastic in the reference (MLXRandom.normal, Layers.swift:274); here the
noise tensors are explicit inputs (one [B, T_i] array per decoder block, or None = zeros).

Arithmetic is float32 throughout (dtype=np.float64 gives a high-precision second opinion).
Layout is the reference's NCT ([batch, channels, time]).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from . import synth


@dataclass
class SnacConfig:
    """SNAC/Config.swift:10-37 (values = hubertsiuzdak/snac_24khz)."""
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

    @property
    def resolved_latent_dim(self) -> int:      # SNACDecoder.swift:50
        return self.latent_dim or self.encoder_dim * 2 ** len(self.encoder_rates)

    @property
    def hop_length(self) -> int:               # SNACDecoder.swift:54
        return int(np.prod(self.encoder_rates))

    def to_json_dict(self) -> dict:
        return dict(sampling_rate=self.sampling_rate, encoder_dim=self.encoder_dim,
                    encoder_rates=list(self.encoder_rates), latent_dim=self.latent_dim,
                    decoder_dim=self.decoder_dim, decoder_rates=list(self.decoder_rates),
                    attn_window_size=self.attn_window_size, codebook_size=self.codebook_size,
                    codebook_dim=self.codebook_dim, vq_strides=list(self.vq_strides),
                    noise=self.noise, depthwise=self.depthwise)


# a 32 / 44 kHz-style configuration in miniature: LocalMHA (window 32, head size 64) after the decoder stem and before the encoder's
# last conv, four codebooks
TINY_ATTN = dict(sampling_rate=32000, encoder_dim=8, encoder_rates=[2, 2, 2, 2], decoder_dim=128, decoder_rates=[2, 2, 2, 2],
                 attn_window_size=32, codebook_size=64, codebook_dim=8, vq_strides=[8, 4, 2, 1], noise=True, depthwise=True)

TINY = dict(encoder_dim=4, encoder_rates=[2, 2, 2, 2], decoder_dim=64, decoder_rates=[4, 2, 2, 2],
            codebook_size=64, codebook_dim=8, vq_strides=[4, 2, 1])


# ----------------------------------------------------------------------------- primitives

def conv1d_nct(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """MLX.conv1d restated on NCT data.  x [B,Cin,T], w [Cout,K,Cin/groups] (MLX layout)."""
    B, Cin, T = x.shape
    Cout, K, Cg = w.shape
    assert Cin == Cg * groups and Cout % groups == 0
    xp = np.zeros((B, Cin, T + 2 * padding), x.dtype)
    xp[:, :, padding:padding + T] = x
    Tout = (T + 2 * padding - dilation * (K - 1) - 1) // stride + 1
    y = np.zeros((B, Cout, Tout), x.dtype)
    og = Cout // groups
    for k in range(K):
        xs = xp[:, :, k * dilation: k * dilation + (Tout - 1) * stride + 1: stride]   # [B,Cin,Tout]
        if groups == 1:
            y += np.einsum("oc,bct->bot", w[:, k, :], xs)
        elif Cg == 1 and og == 1:                                                     # depthwise
            y += w[None, :, k, 0, None] * xs
        else:
            for g in range(groups):
                y[:, g * og:(g + 1) * og] += np.einsum(
                    "oc,bct->bot", w[g * og:(g + 1) * og, k, :], xs[:, g * Cg:(g + 1) * Cg])
    if bias is not None:
        y = y + bias[None, :, None]
    return y


def conv_transpose1d_nct(x, w_in_k_out, bias=None, stride=1, padding=0):
    """WNConvTranspose1d's conv (Layers.swift:166-179).  x [B,Cin,T]; weight in the
    reference's stored layout [Cin, K, Cout] (the swapAxes(0,2) to MLX's [Cout,K,Cin] is
    folded into the indexing).  groups=1, dilation=1 (the only use)."""
    B, Cin, T = x.shape
    Cin2, K, Cout = w_in_k_out.shape
    assert Cin == Cin2
    Tfull = (T - 1) * stride + K
    full = np.zeros((B, Cout, Tfull), x.dtype)
    for k in range(K):
        full[:, :, k: k + (T - 1) * stride + 1: stride] += np.einsum("co,bct->bot", w_in_k_out[:, k, :], x)
    Tout = (T - 1) * stride - 2 * padding + (K - 1) + 1
    y = full[:, :, padding: padding + Tout]
    if bias is not None:
        y = y + bias[None, :, None]
    return y


def wn_conv_weight(g, v):
    """Layers.swift:102-103:  g * v / (||v||_(k,in) + 1e-12);  v [out,K,in/groups], g [out,1,1]."""
    norm = np.sqrt(np.sum(v * v, axis=(1, 2), keepdims=True, dtype=v.dtype))
    return (g * v / (norm + v.dtype.type(1e-12))).astype(v.dtype)


def wn_convT_weight(g, v):
    """Layers.swift:166:  g * v / ||v||_(k,out)  (no eps);  v [in,K,out], g [in,1,1]."""
    norm = np.sqrt(np.sum(v * v, axis=(1, 2), keepdims=True, dtype=v.dtype))
    return (g * v / norm).astype(v.dtype)


def snake(x, alpha):
    """Layers.swift:44-50:  x + 1/(alpha+1e-9) * sin(alpha*x)^2 ; alpha [1,C,1]."""
    t = x.dtype.type
    recip = t(1.0) / (alpha + t(1e-9))
    s = np.sin(alpha * x)
    return x + recip * (s * s)


# ----------------------------------------------------------------------------- model

class SnacOracle:
    def __init__(self, cfg: SnacConfig, weights: dict, dtype=np.float32):
        self.cfg = cfg
        self.dtype = np.dtype(dtype)
        self.w = {k: np.asarray(v, dtype=self.dtype) for k, v in weights.items()}
        # the non-depthwise stem is not on the 24 / 32 / 44 kHz paths (Layers.swift:376-397) and not restated
        assert cfg.depthwise, "non-depthwise decoder stem not restated"

    # -- weight helpers ------------------------------------------------------
    def _wn(self, prefix, bias=True):
        w = wn_conv_weight(self.w[prefix + ".weight_g"], self.w[prefix + ".weight_v"])
        return w, (self.w[prefix + ".bias"] if bias else None)

    # -- quantizer.fromCodes (VQ.swift:165-191) -------------------------------
    def from_codes(self, codes):
        zq = None
        for i, stride in enumerate(self.cfg.vq_strides):
            p = f"quantizer.quantizers.{i}"
            ids = np.asarray(codes[i])
            assert ids.ndim == 2
            zp = self.w[p + ".codebook.weight"][ids]            # [B,T_i,D]   (VQ.swift:88-90)
            zp = np.transpose(zp, (0, 2, 1))                    # [B,D,T_i]   (:92-94)
            w, b = self._wn(p + ".out_proj")
            zqi = conv1d_nct(zp, w, b)                          # (:172)
            if stride > 1:
                zqi = np.repeat(zqi, stride, axis=2)            # (:175-184) repeat_interleave
            zq = zqi if zq is None else zq + zqi                # (:186)  0.0 + z0 + z1 + z2
        return zq.astype(self.dtype)

    # -- LocalMHA (Attention.swift:14-94,99-185): the 32 / 44 kHz models ---------
    def local_mha(self, x, p, dim_head=64):
        """x [B, C, T] -> [B, C, T].  LayerNorm over channels, to_qkv (no bias), attention inside non-overlapping windows of
        attn_window_size frames (full, non-causal), rotary embedding of q and k over the in-window position (SinusoidalEmbeddings
        with useXPos false: scale = 1; freqs = [t*inv_freq, t*inv_freq], rotate_half = [-x2, x1]), scores / sqrt(d), softmax,
        to_out (no bias), + residual."""
        t = self.dtype.type
        win = self.cfg.attn_window_size
        B, C, T = x.shape
        assert T % win == 0, "LocalMHA needs a whole number of windows (the reference's reshape fails otherwise)"
        heads, W = C // dim_head, T // win
        h = np.transpose(x, (0, 2, 1))                                              # [B, T, C]
        mu = h.mean(axis=-1, keepdims=True, dtype=self.dtype)
        var = ((h - mu) ** 2).mean(axis=-1, keepdims=True, dtype=self.dtype)
        h = (h - mu) / np.sqrt(var + t(1e-5)) * self.w[p + ".norm.weight"] + self.w[p + ".norm.bias"]
        qkv = h @ self.w[p + ".to_qkv.weight"].T                                    # Linear(dim, 3 dim, bias: false)
        q, k, v = np.split(qkv, 3, axis=-1)

        def rearr(a):                                                               # "b (w n) (h d) -> b h w n d"
            return np.transpose(a.reshape(B, W, win, heads, dim_head), (0, 3, 1, 2, 4))
        q, k, v = rearr(q), rearr(k), rearr(v)
        inv = self.w.get(p + ".rel_pos.inv_freq")
        if inv is None:
            inv = (t(1.0) / np.power(t(10000.0), np.arange(0, dim_head, 2, dtype=self.dtype) / t(dim_head))).astype(self.dtype)
        pos = np.arange(win, dtype=self.dtype)
        fr = pos[:, None] * inv[None, :]
        fr = np.concatenate([fr, fr], axis=-1)                                      # [n, d]
        cos, sin = np.cos(fr).astype(self.dtype), np.sin(fr).astype(self.dtype)

        def rot_half(a):
            a1, a2 = a[..., : dim_head // 2], a[..., dim_head // 2:]
            return np.concatenate([-a2, a1], axis=-1)
        q = q * cos + rot_half(q) * sin
        k = k * cos + rot_half(k) * sin
        sc = (q @ np.swapaxes(k, -1, -2)) / t(math.sqrt(dim_head))
        sc = sc - sc.max(axis=-1, keepdims=True)
        pr = np.exp(sc)
        pr = pr / pr.sum(axis=-1, keepdims=True, dtype=self.dtype)
        o = pr @ v                                                                  # [B, h, w, n, d]
        o = np.transpose(o, (0, 2, 3, 1, 4)).reshape(B, T, C)                       # "b h w n d -> b (w n) (h d)"
        o = o @ self.w[p + ".to_out.weight"].T
        return (np.transpose(o, (0, 2, 1)) + x).astype(self.dtype)

    # -- decoder (Layers.swift:364-421) ---------------------------------------
    def _residual_unit(self, x, p, dilation):
        # Layers.swift:202-231 ; groups = dim (depthwise), kernel 7, pad 3*dil
        h = snake(x, self.w[p + ".block.layers.0.alpha"])
        w, b = self._wn(p + ".block.layers.1")
        h = conv1d_nct(h, w, b, padding=3 * dilation, dilation=dilation, groups=x.shape[1])
        h = snake(h, self.w[p + ".block.layers.2.alpha"])
        w, b = self._wn(p + ".block.layers.3")
        h = conv1d_nct(h, w, b)
        return x + h                                            # (:230) same length => no crop

    def _decoder_block(self, x, p, stride, noise):
        # Layers.swift:286-309
        x = snake(x, self.w[p + ".block.layers.0.alpha"])
        wt = wn_convT_weight(self.w[p + ".block.layers.1.weight_g"], self.w[p + ".block.layers.1.weight_v"])
        x = conv_transpose1d_nct(x, wt, self.w[p + ".block.layers.1.bias"], stride=stride,
                                 padding=int(math.ceil(stride / 2.0)))
        idx = 2
        if self.cfg.noise:
            w, _ = self._wn(p + ".block.layers.2.linear", bias=False)
            h = conv1d_nct(x, w, None)                          # Layers.swift:275
            if noise is not None:
                x = x + noise[:, None, :].astype(self.dtype) * h   # (:276-277)
            # noise None == zeros: x + 0*h == x exactly for finite h
            idx = 3
        for j, dil in enumerate((1, 3, 9)):
            x = self._residual_unit(x, f"{p}.block.layers.{idx + j}", dil)
        return x

    def decoder(self, zq, noises=None, return_intermediates=False):
        cfg = self.cfg
        inter = {}
        p = "decoder.model.layers"
        w, b = self._wn(p + ".0")
        x = conv1d_nct(zq, w, b, padding=3, groups=zq.shape[1])            # Layers.swift:380-386
        inter["stem_dw"] = x
        w, b = self._wn(p + ".1")
        x = conv1d_nct(x, w, b)                                            # :387
        inter["stem_pw"] = x
        first = 2
        if cfg.attn_window_size:                                           # :395-397
            x = self.local_mha(x, p + ".2")
            inter["attn"] = x
            first = 3
        for i, s in enumerate(cfg.decoder_rates):
            nz = None if noises is None else noises[i]
            x = self._decoder_block(x, f"{p}.{first + i}", s, nz)          # :399-405
            inter[f"block{i}"] = x
        n = first + len(cfg.decoder_rates)
        x = snake(x, self.w[f"{p}.{n}.alpha"])                             # :410
        w, b = self._wn(f"{p}.{n + 1}")
        x = conv1d_nct(x, w, b, padding=3)                                 # :411
        x = np.tanh(x)                                                     # :412
        return (x, inter) if return_intermediates else x

    def decode(self, codes, noises=None):
        """SNAC.decode (SNACDecoder.swift:127-131): codes = 3 int arrays [B,T_i] -> [B,1,N]."""
        return self.decoder(self.from_codes(codes), noises)

    # -- encoder + RVQ encode (Layers.swift:236-259,319-360; VQ.swift:47-163; SNACDecoder.swift:86-125) ----
    def preprocess(self, audio):
        """Right-pad to a multiple of hop_length * lcm(vq_strides [, attn_window_size]) (SNACDecoder.swift:86-104)."""
        cfg = self.cfg
        l = 1
        for s in cfg.vq_strides:
            l = l * s // math.gcd(l, s)
        if cfg.attn_window_size:
            l = l * cfg.attn_window_size // math.gcd(l, cfg.attn_window_size)
        pad_to = cfg.hop_length * l
        n = audio.shape[-1]
        right = int(math.ceil(n / pad_to)) * pad_to - n
        return np.pad(audio, [(0, 0)] * (audio.ndim - 1) + [(0, right)])

    def encoder(self, audio):
        """audio [B, 1, T] -> z [B, latent, T / hop]."""
        cfg = self.cfg
        p = "encoder.block.layers"
        w, b = self._wn(p + ".0")
        x = conv1d_nct(np.asarray(audio, self.dtype), w, b, padding=3)
        for i, s in enumerate(cfg.encoder_rates):
            q = f"{p}.{1 + i}.block.layers"
            for j, dil in enumerate((1, 3, 9)):
                x = self._residual_unit(x, f"{q}.{j}", dil) if cfg.depthwise else self._residual_unit_dense(x, f"{q}.{j}", dil)
            x = snake(x, self.w[q + ".3.alpha"])
            w, b = self._wn(q + ".4")
            x = conv1d_nct(x, w, b, stride=s, padding=int(math.ceil(s / 2.0)))
        n = 1 + len(cfg.encoder_rates)
        if cfg.attn_window_size:                                           # Layers.swift:339-341
            x = self.local_mha(x, f"{p}.{n}")
            n += 1
        w, b = self._wn(f"{p}.{n}")
        return conv1d_nct(x, w, b, padding=3, groups=x.shape[1] if cfg.depthwise else 1)

    def quantize(self, z, return_details=False):
        """ResidualVectorQuantize.callAsFunction (VQ.swift:141-161): codes per level + per-level distance tables (details)."""
        t = self.dtype.type
        residual = np.array(z, self.dtype)
        codes, details = [], []
        for i, stride in enumerate(self.cfg.vq_strides):
            p = f"quantizer.quantizers.{i}"
            r = residual
            if stride > 1:                                                   # avg pool, kernel = stride (:47-57)
                B, C, T = r.shape
                r = r[:, :, : (T // stride) * stride].reshape(B, C, T // stride, stride).sum(-1, dtype=self.dtype) / t(stride)
            w, b = self._wn(p + ".in_proj")
            ze = conv1d_nct(r.astype(self.dtype), w, b)                      # [B, 8, T_i]
            B, D, Ti = ze.shape
            e = np.transpose(ze, (0, 2, 1)).reshape(B * Ti, D)
            cb = self.w[p + ".codebook.weight"]
            en = e / np.maximum(np.sqrt((e * e).sum(1, keepdims=True)), t(1e-12))
            cn = cb / np.maximum(np.sqrt((cb * cb).sum(1, keepdims=True)), t(1e-12))
            dist = (en * en).sum(1, keepdims=True) - t(2.0) * (en @ cn.T) + (cn * cn).sum(1, keepdims=True).T
            idx = np.argmax(-dist, axis=1).reshape(B, Ti)                    # first index on ties
            codes.append(idx.astype(np.int32))
            details.append(dist.reshape(B, Ti, -1))
            zq = np.transpose(cb[idx], (0, 2, 1))
            w, b = self._wn(p + ".out_proj")
            zqi = conv1d_nct(zq.astype(self.dtype), w, b)
            if stride > 1:
                zqi = np.repeat(zqi, stride, axis=2)
            residual = (residual - zqi).astype(self.dtype)
        return (codes, details) if return_details else codes

    def encode(self, audio, return_details=False):
        """SNAC.encode (SNACDecoder.swift:120-125): audio [B, 1, T] -> [codes_i [B, T_i]]."""
        return self.quantize(self.encoder(self.preprocess(np.asarray(audio, self.dtype))), return_details)

    def noise_lengths(self, groups: int):
        """Time length of each NoiseBlock's input for `groups` Orpheus frames
        (T0 = 4*groups... in general lcm-based; here vq_strides[0]*groups)."""
        t = self.cfg.vq_strides[0] * groups
        out = []
        for s in self.cfg.decoder_rates:
            pad = int(math.ceil(s / 2.0))
            t = (t - 1) * s - 2 * pad + 2 * s
            out.append(t)
        return out


# ----------------------------------------------------------------------------- synthetic weights

def make_synthetic_weights(cfg: SnacConfig, seed: int = 1234, with_encoder: bool = False) -> dict:
    """Seeded synthetic weights in the reference's safetensors key layout,
    weight_v ~ U(+-gain*sqrt(3/fan_in)) (variance-preserving, so the random network keeps O(1)
    activations and the final tanh is exercised off saturation), weight_g = ||v|| * U(0.5,1.5),
    bias ~ U(+-0.1), alpha ~ U(0.5,2),
    codebooks ~ U(-sqrt3, sqrt3) (unit variance).  Decoder + quantizer keys only (the encode
    path is out of this round's scope)."""
    W = {}
    key = [seed * 1000]

    def nxt():
        key[0] += 1
        return key[0]

    def wn_conv(prefix, cout, k, cin_g, cin_total, bias=True, gain=1.0):
        del cin_total                              # reference init uses 1/(in*k); we keep unit gain
        amp = gain * math.sqrt(3.0 / (cin_g * k))  # so activations stay O(1) through the stack
        v = synth.synth_tensor(nxt(), (cout, k, cin_g), amp)
        nrm = np.sqrt(np.sum(v * v, axis=(1, 2), keepdims=True))
        g = nrm * (1.0 + synth.synth_tensor(nxt(), (cout, 1, 1), 0.5))
        W[prefix + ".weight_v"] = v.astype(np.float32)
        W[prefix + ".weight_g"] = g.astype(np.float32)
        if bias:
            W[prefix + ".bias"] = synth.synth_tensor(nxt(), (cout,), 0.1)

    def alpha(name, c):
        W[name] = (1.25 + synth.synth_tensor(nxt(), (1, c, 1), 0.75)).astype(np.float32)

    D = cfg.resolved_latent_dim
    for i in range(len(cfg.vq_strides)):
        p = f"quantizer.quantizers.{i}"
        W[p + ".codebook.weight"] = synth.synth_tensor(nxt(), (cfg.codebook_size, cfg.codebook_dim), math.sqrt(3.0))
        wn_conv(p + ".in_proj", cfg.codebook_dim, 1, D, D)
        wn_conv(p + ".out_proj", D, 1, cfg.codebook_dim, cfg.codebook_dim)
    p = "decoder.model.layers"
    wn_conv(p + ".0", D, 7, 1, D)                 # depthwise: reference init scale uses inChannels*k
    wn_conv(p + ".1", cfg.decoder_dim, 1, D, D)
    first = 3 if cfg.attn_window_size else 2

    def mha(prefix, dim):                         # LocalMHA parameters (Attention.swift:22-29)
        W[prefix + ".norm.weight"] = (1.0 + synth.synth_tensor(nxt(), (dim,), 0.2)).astype(np.float32)
        W[prefix + ".norm.bias"] = synth.synth_tensor(nxt(), (dim,), 0.1)
        W[prefix + ".to_qkv.weight"] = synth.synth_tensor(nxt(), (3 * dim, dim), math.sqrt(3.0 / dim))
        W[prefix + ".to_out.weight"] = synth.synth_tensor(nxt(), (dim, dim), 0.5 * math.sqrt(3.0 / dim))
        W[prefix + ".rel_pos.inv_freq"] = (1.0 / np.power(10000.0, np.arange(0, 64, 2, dtype=np.float32) / 64.0)).astype(np.float32)
    for i, s in enumerate(cfg.decoder_rates):
        cin = cfg.decoder_dim // 2 ** i
        cout = cfg.decoder_dim // 2 ** (i + 1)
        b = f"{p}.{first + i}.block.layers"
        alpha(b + ".0.alpha", cin)
        amp = math.sqrt(3.0 / (cin * 2))          # two taps of each input reach one output
        v = synth.synth_tensor(nxt(), (cin, 2 * s, cout), amp)
        nrm = np.sqrt(np.sum(v * v, axis=(1, 2), keepdims=True))
        W[b + ".1.weight_v"] = v.astype(np.float32)
        W[b + ".1.weight_g"] = (nrm * (1.0 + synth.synth_tensor(nxt(), (cin, 1, 1), 0.5))).astype(np.float32)
        W[b + ".1.bias"] = synth.synth_tensor(nxt(), (cout,), 0.1)
        idx = 2
        if cfg.noise:
            wn_conv(b + ".2.linear", cout, 1, cout, cout, bias=False, gain=0.3)
            idx = 3
        for j in range(3):
            r = f"{b}.{idx + j}.block.layers"
            alpha(r + ".0.alpha", cout)
            wn_conv(r + ".1", cout, 7, 1, cout)
            alpha(r + ".2.alpha", cout)
            wn_conv(r + ".3", cout, 1, cout, cout, gain=0.2)
    n = first + len(cfg.decoder_rates)
    cl = cfg.decoder_dim // 2 ** len(cfg.decoder_rates)
    alpha(f"{p}.{n}.alpha", cl)
    wn_conv(f"{p}.{n + 1}", 1, 7, cl, cl, gain=0.12)
    if cfg.attn_window_size:                      # after the other decoder keys: they keep their generator keys
        mha(p + ".2", cfg.decoder_dim)
    if with_encoder:                      # appended after every decoder key: the decoder tensors keep their generator keys
        e = "encoder.block.layers"
        wn_conv(e + ".0", cfg.encoder_dim, 7, 1, 1, gain=3.0)
        c = cfg.encoder_dim
        for i, st in enumerate(cfg.encoder_rates):
            b = f"{e}.{1 + i}.block.layers"
            for j in range(3):
                r = f"{b}.{j}.block.layers"
                alpha(r + ".0.alpha", c)
                wn_conv(r + ".1", c, 7, 1 if cfg.depthwise else c, c)
                alpha(r + ".2.alpha", c)
                wn_conv(r + ".3", c, 1, c, c, gain=0.2)
            alpha(b + ".3.alpha", c)
            wn_conv(b + ".4", 2 * c, 2 * st, c, c)
            c *= 2
        ne = 1 + len(cfg.encoder_rates)
        if cfg.attn_window_size:
            mha(f"{e}.{ne}", c)
            ne += 1
        wn_conv(f"{e}.{ne}", c, 7, 1 if cfg.depthwise else c, c)
    return W


def synthetic_codes(cfg: SnacConfig, batch: int, groups: int, seed: int = 1235):
    """Uniform random codebook indices, one array per VQ level: [B, groups * s0/stride_i]."""
    out = []
    s0 = cfg.vq_strides[0]
    for i, s in enumerate(cfg.vq_strides):
        n = batch * groups * (s0 // s)
        u = synth.uniform01(seed * 100 + i, n)
        out.append(np.minimum((u * cfg.codebook_size).astype(np.int32), cfg.codebook_size - 1)
                   .reshape(batch, groups * (s0 // s)))
    return out


def synthetic_noise(cfg: SnacConfig, batch: int, groups: int, seed: int = 1236):
    """Approximately N(0,1) noise (sum of 12 uniforms - 6), one [B, T_i] array per block."""
    o = SnacOracle.__new__(SnacOracle)
    o.cfg = cfg
    outs = []
    for i, t in enumerate(o.noise_lengths(groups)):
        n = batch * t
        acc = np.zeros(n, np.float32)
        for j in range(12):
            acc += synth.uniform01(seed * 1000 + i * 16 + j, n)
        outs.append((acc - np.float32(6.0)).reshape(batch, t))
    return outs
