"""Seeded synthetic SNAC weights for benches and smoke runs (there are no checkpoints offline).
Generator "mis-synth-v1": element i of the tensor with 64-bit key k is
    u = splitmix64(k * 0x9E3779B97F4A7C15 + i);  x = ((u >> 40) + 0.5) * 2^-24;  value = (2x - 1) * amp
(the LM uses the same generator on the device, csrc/lm_kernels.hip k_synth_fill_bf16).
Key layout = the reference's safetensors names; decoder + quantizer only."""
from __future__ import annotations

import math

import numpy as np


def _splitmix64(z):
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def synth_tensor(key: int, shape, amplitude: float) -> np.ndarray:
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(key) * np.uint64(0x9E3779B97F4A7C15)
    u = _splitmix64(idx)
    x = ((u >> np.uint64(40)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)
    return ((np.float32(2.0) * x - np.float32(1.0)) * np.float32(amplitude)).reshape(shape)


def snac_synthetic_weights(cfg, seed: int = 1234) -> dict:
    """cfg: codecs.SNACConfig.  Variance-preserving uniform weights, weight_g = ||v|| * U(0.5,1.5),
    bias U(+-0.1), alpha U(0.5,2), codebooks unit variance."""
    W = {}
    key = [seed * 1000]

    def nxt():
        key[0] += 1
        return key[0]

    def wn_conv(prefix, cout, k, cin_g, bias=True, gain=1.0):
        amp = gain * math.sqrt(3.0 / (cin_g * k))
        v = synth_tensor(nxt(), (cout, k, cin_g), amp)
        nrm = np.sqrt(np.sum(v * v, axis=(1, 2), keepdims=True))
        g = nrm * (1.0 + synth_tensor(nxt(), (cout, 1, 1), 0.5))
        W[prefix + ".weight_v"] = v.astype(np.float32)
        W[prefix + ".weight_g"] = g.astype(np.float32)
        if bias:
            W[prefix + ".bias"] = synth_tensor(nxt(), (cout,), 0.1)

    def alpha(name, c):
        W[name] = (1.25 + synth_tensor(nxt(), (1, c, 1), 0.75)).astype(np.float32)

    D = cfg.latent_dim or cfg.encoder_dim * 2 ** len(cfg.encoder_rates)
    for i in range(len(cfg.vq_strides)):
        p = f"quantizer.quantizers.{i}"
        W[p + ".codebook.weight"] = synth_tensor(nxt(), (cfg.codebook_size, cfg.codebook_dim), math.sqrt(3.0))
        wn_conv(p + ".in_proj", cfg.codebook_dim, 1, D)
        wn_conv(p + ".out_proj", D, 1, cfg.codebook_dim)
    p = "decoder.model.layers"
    wn_conv(p + ".0", D, 7, 1)
    wn_conv(p + ".1", cfg.decoder_dim, 1, D)
    for i, s in enumerate(cfg.decoder_rates):
        cin = cfg.decoder_dim // 2 ** i
        cout = cfg.decoder_dim // 2 ** (i + 1)
        b = f"{p}.{2 + i}.block.layers"
        alpha(b + ".0.alpha", cin)
        v = synth_tensor(nxt(), (cin, 2 * s, cout), math.sqrt(3.0 / (cin * 2)))
        nrm = np.sqrt(np.sum(v * v, axis=(1, 2), keepdims=True))
        W[b + ".1.weight_v"] = v.astype(np.float32)
        W[b + ".1.weight_g"] = (nrm * (1.0 + synth_tensor(nxt(), (cin, 1, 1), 0.5))).astype(np.float32)
        W[b + ".1.bias"] = synth_tensor(nxt(), (cout,), 0.1)
        idx = 2
        if cfg.noise:
            wn_conv(b + ".2.linear", cout, 1, cout, bias=False, gain=0.3)
            idx = 3
        for j in range(3):
            r = f"{b}.{idx + j}.block.layers"
            alpha(r + ".0.alpha", cout)
            wn_conv(r + ".1", cout, 7, 1)
            alpha(r + ".2.alpha", cout)
            wn_conv(r + ".3", cout, 1, cout, gain=0.2)
    n = 2 + len(cfg.decoder_rates)
    cl = cfg.decoder_dim // 2 ** len(cfg.decoder_rates)
    alpha(f"{p}.{n}.alpha", cl)
    wn_conv(f"{p}.{n + 1}", 1, 7, cl, gain=0.12)
    return W


def soprano_decoder_synthetic_weights(cfg, seed: int = 99) -> dict:
    """cfg: soprano.SopranoConfiguration.  Keys as stored in the checkpoint ("decoder.*", float32)."""
    W, key = {}, [seed * 100000]

    def t(shape, amp):
        key[0] += 1
        return synth_tensor(key[0], shape, amp)

    F = np.float32
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
