"""Pin the SNAC oracle against an independent implementation (torch.nn.functional) of the
conv definitions the reference relies on, and against float64."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

from oracle import snac


@pytest.mark.parametrize("groups,dil,k,stride,pad", [(1, 1, 7, 1, 3), (8, 3, 7, 1, 9), (8, 9, 7, 1, 27),
                                                     (1, 1, 1, 1, 0), (2, 1, 4, 2, 1), (1, 1, 16, 8, 4)])
def test_conv1d_matches_torch(groups, dil, k, stride, pad):
    rng = np.random.default_rng(0)
    cin, cout = 8, 8
    x = rng.standard_normal((2, cin, 50)).astype(np.float32)
    w = rng.standard_normal((cout, k, cin // groups)).astype(np.float32)      # MLX layout
    b = rng.standard_normal(cout).astype(np.float32)
    y = snac.conv1d_nct(x, w, b, stride=stride, padding=pad, dilation=dil, groups=groups)
    yt = Fn.conv1d(torch.from_numpy(x), torch.from_numpy(w).permute(0, 2, 1).contiguous(), torch.from_numpy(b),
                   stride=stride, padding=pad, dilation=dil, groups=groups).numpy()
    assert y.shape == yt.shape
    np.testing.assert_allclose(y, yt, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("stride", [2, 3, 4, 7, 8])
def test_conv_transpose_matches_torch_and_length(stride):
    rng = np.random.default_rng(1)
    cin, cout, T = 6, 4, 11
    k, pad = 2 * stride, int(math.ceil(stride / 2))
    x = rng.standard_normal((2, cin, T)).astype(np.float32)
    w = rng.standard_normal((cin, k, cout)).astype(np.float32)                # reference layout [in,k,out]
    b = rng.standard_normal(cout).astype(np.float32)
    y = snac.conv_transpose1d_nct(x, w, b, stride=stride, padding=pad)
    yt = Fn.conv_transpose1d(torch.from_numpy(x), torch.from_numpy(w).permute(0, 2, 1).contiguous(),
                             torch.from_numpy(b), stride=stride, padding=pad).numpy()
    np.testing.assert_allclose(y, yt, rtol=1e-4, atol=1e-4)
    # s*T for even s, s*T - 1 for odd s (output_padding dropped, Layers.swift:169-176)
    assert y.shape[2] == (stride * T if stride % 2 == 0 else stride * T - 1)


def _torch_decode(cfg, W, codes, noises):
    """Independent re-implementation of the module tree with torch functional ops (float64)."""
    t = lambda a: torch.from_numpy(np.asarray(a, np.float64))

    def wn(p):
        v, g = t(W[p + ".weight_v"]), t(W[p + ".weight_g"])
        return g * v / (v.pow(2).sum((1, 2), keepdim=True).sqrt() + 1e-12)

    def conv(x, p, bias=True, **kw):
        return Fn.conv1d(x, wn(p).permute(0, 2, 1).contiguous(), t(W[p + ".bias"]) if bias else None, **kw)

    def snk(x, a):
        a = t(W[a])
        return x + (1.0 / (a + 1e-9)) * torch.sin(a * x) ** 2

    zq = 0
    for i, s in enumerate(cfg.vq_strides):
        p = f"quantizer.quantizers.{i}"
        z = t(W[p + ".codebook.weight"])[torch.from_numpy(codes[i]).long()].transpose(1, 2)
        z = conv(z, p + ".out_proj")
        zq = zq + z.repeat_interleave(s, dim=2)
    p = "decoder.model.layers"
    x = conv(zq, p + ".0", padding=3, groups=zq.shape[1])
    x = conv(x, p + ".1")
    for i, s in enumerate(cfg.decoder_rates):
        b = f"{p}.{2 + i}.block.layers"
        x = snk(x, b + ".0.alpha")
        v, g = t(W[b + ".1.weight_v"]), t(W[b + ".1.weight_g"])
        wt = g * v / v.pow(2).sum((1, 2), keepdim=True).sqrt()
        x = Fn.conv_transpose1d(x, wt.permute(0, 2, 1).contiguous(), t(W[b + ".1.bias"]), stride=s,
                                padding=int(math.ceil(s / 2)))
        x = x + t(noises[i])[:, None, :] * conv(x, b + ".2.linear", bias=False)
        for j, d in enumerate((1, 3, 9)):
            r = f"{b}.{3 + j}.block.layers"
            h = snk(x, r + ".0.alpha")
            h = conv(h, r + ".1", padding=3 * d, dilation=d, groups=x.shape[1])
            h = snk(h, r + ".2.alpha")
            x = x + conv(h, r + ".3")
    n = 2 + len(cfg.decoder_rates)
    x = snk(x, f"{p}.{n}.alpha")
    return torch.tanh(conv(x, f"{p}.{n + 1}", padding=3)).numpy()


@pytest.mark.parametrize("cfgd,batch,groups", [(snac.TINY, 2, 5), ({}, 1, 2)])
def test_full_decode_matches_independent_torch_float64(cfgd, batch, groups):
    cfg = snac.SnacConfig(**cfgd)
    W = snac.make_synthetic_weights(cfg)
    codes = snac.synthetic_codes(cfg, batch, groups)
    nz = snac.synthetic_noise(cfg, batch, groups)
    y = snac.SnacOracle(cfg, W).decode(codes, nz)
    assert y.shape == (batch, 1, groups * cfg.vq_strides[0] * int(np.prod(cfg.decoder_rates)))
    ref = _torch_decode(cfg, W, codes, nz)
    rms = float(np.sqrt(np.mean((y - ref) ** 2)))
    assert rms < 2e-5, rms
    assert np.all(np.abs(y) < 1.0)
    # decoded waveform must not be degenerate (tanh neither dead nor saturated)
    assert 0.05 < y.std() < 0.9


def test_c1_shape_and_zero_noise_equals_no_noise():
    # BASELINE config 1: G=12 -> 24576 samples; run on the tiny net for speed
    cfg = snac.SnacConfig(**snac.TINY)
    W = snac.make_synthetic_weights(cfg)
    o = snac.SnacOracle(cfg, W)
    codes = snac.synthetic_codes(cfg, 1, 12)
    zeros = [np.zeros((1, t), np.float32) for t in o.noise_lengths(12)]
    a = o.decode(codes, None)
    b = o.decode(codes, zeros)
    assert a.shape == (1, 1, 12 * 4 * 32)
    assert np.array_equal(a, b)
    full = snac.SnacConfig()
    assert full.resolved_latent_dim == 768 and full.hop_length == 512
    assert snac.SnacOracle(full, {}).noise_lengths(12) == [384, 3072, 12288, 24576]


def test_encode_path_strided_conv_and_nearest_code():
    import torch
    import torch.nn.functional as TF
    cfg = snac.SnacConfig(**snac.TINY)
    W = snac.make_synthetic_weights(cfg, with_encoder=True)
    W0 = snac.make_synthetic_weights(cfg)
    assert all(np.array_equal(W[k], W0[k]) for k in W0)               # decoder tensors keep their generator keys
    orc = snac.SnacOracle(cfg, W)
    rng = np.random.default_rng(3)
    # strided WNConv1d against torch
    x = rng.standard_normal((2, 8, 40)).astype(np.float32)
    w = rng.standard_normal((16, 4, 8)).astype(np.float32)
    ref = TF.conv1d(torch.from_numpy(x), torch.from_numpy(w).permute(0, 2, 1), stride=2, padding=1).numpy()
    assert np.abs(snac.conv1d_nct(x, w, stride=2, padding=1) - ref).max() < 1e-5
    # preprocess pads to hop * lcm(vq_strides)
    audio = (0.3 * rng.standard_normal((2, 1, 1000))).astype(np.float32)
    assert orc.preprocess(audio).shape[-1] == 1024 and cfg.hop_length * 4 == 64
    codes, details = orc.encode(audio, return_details=True)
    assert [c.shape for c in codes] == [(2, 16), (2, 32), (2, 64)]
    # nearest code == cosine-similarity argmax in float64
    z = orc.encoder(orc.preprocess(audio)).astype(np.float64)
    pooled = z.reshape(2, z.shape[1], 16, 4).mean(-1)
    wi, bi = orc._wn("quantizer.quantizers.0.in_proj")
    e = np.einsum("oc,bct->bto", wi[:, 0, :].astype(np.float64), pooled) + bi
    cb = W["quantizer.quantizers.0.codebook.weight"].astype(np.float64)
    cos = (e / np.linalg.norm(e, axis=-1, keepdims=True)) @ (cb / np.linalg.norm(cb, axis=-1, keepdims=True)).T
    best = cos.argmax(-1)
    agree = (best == codes[0])
    margin = np.sort(cos, -1)[..., -1] - np.sort(cos, -1)[..., -2]
    assert agree[margin > 1e-5].all() and agree.mean() > 0.9
    # codes decode back through fromCodes to the sum of the per-level quantised vectors
    zq = orc.from_codes(codes)
    assert zq.shape == z.shape


def test_local_mha_matches_an_independent_torch_implementation():
    """LocalMHA of the 32 / 44 kHz models (Attention.swift:14-185) against torch building blocks (nn.LayerNorm, F.linear,
    F.scaled_dot_product_attention per window, rotary embedding written the complex-pair way)."""
    import torch
    import torch.nn.functional as F
    from oracle import snac as osn
    cfg = osn.SnacConfig(**osn.TINY_ATTN)
    W = osn.make_synthetic_weights(cfg, with_encoder=True)
    o = osn.SnacOracle(cfg, W)
    p = "decoder.model.layers.2"
    rng = np.random.default_rng(0)
    B, C, T = 2, cfg.decoder_dim, 96
    x = rng.standard_normal((B, C, T)).astype(np.float32)
    got = o.local_mha(x, p)
    xt = torch.from_numpy(x).transpose(1, 2)                                         # [B, T, C]
    h = F.layer_norm(xt, (C,), torch.from_numpy(W[p + ".norm.weight"]), torch.from_numpy(W[p + ".norm.bias"]), 1e-5)
    qkv = F.linear(h, torch.from_numpy(W[p + ".to_qkv.weight"]))
    q, k, v = qkv.chunk(3, dim=-1)
    win, heads = 32, C // 64

    def windows(a):                                                                  # [B, T, C] -> [B, heads, T/win, win, 64]
        return a.reshape(B, T // win, win, heads, 64).permute(0, 3, 1, 2, 4)
    q, k, v = windows(q), windows(k), windows(v)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2).float() / 64))
    ang = torch.arange(win).float()[:, None] * inv[None, :]                          # [win, 32]
    rot = torch.polar(torch.ones_like(ang), ang)                                     # e^{i n f_j}: pairs (j, j + 32) as one complex number

    def rope(a):
        z = torch.complex(a[..., :32], a[..., 32:]) * rot
        return torch.cat([z.real, z.imag], dim=-1)
    out = F.scaled_dot_product_attention(rope(q), rope(k), v)                        # scale 1 / sqrt(64), full attention inside the window
    out = out.permute(0, 2, 3, 1, 4).reshape(B, T, C)
    ref = (F.linear(out, torch.from_numpy(W[p + ".to_out.weight"])).transpose(1, 2) + torch.from_numpy(x)).numpy()
    assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-5 * np.abs(ref).max()
    # a window is self-contained: changing frames of window 2 leaves windows 0 and 1 alone
    x2 = x.copy(); x2[:, :, 64:] += 1.0
    assert np.array_equal(o.local_mha(x2, p)[:, :, :64], got[:, :, :64])


def test_attention_variant_decode_and_encode_shapes():
    from oracle import snac as osn
    cfg = osn.SnacConfig(**osn.TINY_ATTN)
    o = osn.SnacOracle(cfg, osn.make_synthetic_weights(cfg, with_encoder=True))
    codes = osn.synthetic_codes(cfg, 2, 8)                                           # 8 coarse frames -> 64 latent frames = 2 windows
    wav = o.decode(codes, osn.synthetic_noise(cfg, 2, 8))
    assert wav.shape == (2, 1, 64 * 16) and np.abs(wav).max() <= 1.0
    audio = (0.3 * np.random.default_rng(1).standard_normal((1, 1, 1000))).astype(np.float32)
    pa = o.preprocess(audio)
    assert pa.shape[-1] % (16 * 32) == 0                                             # hop 16, lcm(vq stride 8, window 32) = 32
    c = o.encode(audio)
    assert [ci.shape for ci in c] == [(1, pa.shape[-1] // 16 // s) for s in cfg.vq_strides]
