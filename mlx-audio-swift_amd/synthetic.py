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


def qwen3tts_synthetic_weights(cfg, seed: int = 515):
    """cfg: qwen3tts.Qwen3TTSConfiguration.  Yields (name, array) pairs (bf16-valued float32 for the LMs would double the host
    footprint, so LM tensors are produced as float32 and converted by the loader; decoder tensors float32)."""
    key = [seed * 100000]

    def t(shape, amp, offset=0.0):
        key[0] += 1
        a = synth_tensor(key[0], shape, amp)
        return a + np.float32(offset) if offset else a

    def lm(prefix, c):
        d, ff, H, Hkv, D = c.hidden_size, c.intermediate_size, c.num_attention_heads, c.num_key_value_heads, c.head_dim
        yield prefix + "model.norm.weight", t((d,), 0.1, 1.0)
        for li in range(c.num_hidden_layers):
            p = f"{prefix}model.layers.{li}"
            yield p + ".input_layernorm.weight", t((d,), 0.1, 1.0)
            yield p + ".post_attention_layernorm.weight", t((d,), 0.1, 1.0)
            yield p + ".self_attn.q_proj.weight", t((H * D, d), math.sqrt(3.0 / d) * 1.5)
            yield p + ".self_attn.k_proj.weight", t((Hkv * D, d), math.sqrt(3.0 / d) * 1.5)
            yield p + ".self_attn.v_proj.weight", t((Hkv * D, d), math.sqrt(3.0 / d))
            yield p + ".self_attn.o_proj.weight", t((d, H * D), math.sqrt(3.0 / (H * D)) * 0.5)
            yield p + ".mlp.gate_proj.weight", t((ff, d), math.sqrt(3.0 / d))
            yield p + ".mlp.up_proj.weight", t((ff, d), math.sqrt(3.0 / d))
            yield p + ".mlp.down_proj.weight", t((d, ff), math.sqrt(3.0 / ff) * 0.5)
            yield p + ".self_attn.q_norm.weight", t((D,), 0.1, 1.0)
            yield p + ".self_attn.k_norm.weight", t((D,), 0.1, 1.0)

    tk, pr, dc = cfg.talker, cfg.predictor, cfg.decoder
    d, dp, th = tk.hidden_size, pr.hidden_size, cfg.text_hidden_size
    yield "model.codec_embedding.weight", t((tk.vocab_size, d), 0.5 * math.sqrt(3.0))
    yield "codec_head.weight", t((tk.vocab_size, d), math.sqrt(3.0 / d) * 2.0)
    yield "model.text_embedding.weight", t((cfg.text_vocab_size, th), 0.5 * math.sqrt(3.0))
    yield "text_projection.linear_fc1.weight", t((th, th), math.sqrt(3.0 / th) * 2.0)
    yield "text_projection.linear_fc1.bias", t((th,), 0.1)
    yield "text_projection.linear_fc2.weight", t((d, th), math.sqrt(3.0 / th) * 2.0)
    yield "text_projection.linear_fc2.bias", t((d,), 0.1)
    yield from lm("", tk)
    yield from lm("code_predictor.", pr)
    for i in range(cfg.num_code_groups - 1):
        yield f"code_predictor.model.codec_embedding.{i}.weight", t((pr.vocab_size, d), 0.5 * math.sqrt(3.0))
        yield f"code_predictor.lm_head.{i}.weight", t((pr.vocab_size, dp), math.sqrt(3.0 / dp) * 2.0)
    if d != dp:
        yield "code_predictor.small_to_mtp_projection.weight", t((dp, d), math.sqrt(3.0 / d))
        yield "code_predictor.small_to_mtp_projection.bias", t((dp,), 0.1)
    # speech-tokenizer decoder (float32)
    half = dc.codebook_dim // 2
    for name, n in (("rvq_first", dc.num_semantic_quantizers), ("rvq_rest", dc.num_quantizers - dc.num_semantic_quantizers)):
        for i in range(n):
            p = f"decoder.quantizer.{name}.vq.layers.{i}.codebook"
            yield p + ".cluster_usage", t((dc.codebook_size,), 0.5, 1.0)
            yield p + ".embedding_sum", t((dc.codebook_size, half), math.sqrt(3.0) / math.sqrt(n))
        yield f"decoder.quantizer.{name}.output_proj.weight", t((dc.codebook_dim, 1, half), math.sqrt(3.0 / half))

    def conv(p, co, k, ci, gain=1.0):
        yield p + ".weight", t((co, k, ci), gain * math.sqrt(3.0 / (k * ci)))
        yield p + ".bias", t((co,), 0.05)

    def lin(p, co, ci, bias=True, gain=1.0):
        yield p + ".weight", t((co, ci), gain * math.sqrt(3.0 / ci))
        if bias:
            yield p + ".bias", t((co,), 0.05)

    yield from conv("decoder.pre_conv.conv", dc.latent_dim, 3, dc.codebook_dim)
    P, hs = "decoder.pre_transformer", dc.hidden_size
    yield from lin(P + ".input_proj", hs, dc.latent_dim)
    yield from lin(P + ".output_proj", dc.latent_dim, hs)
    yield P + ".norm.weight", t((hs,), 0.1, 1.0)
    for i in range(dc.num_hidden_layers):
        p = f"{P}.layers.{i}"
        yield p + ".input_layernorm.weight", t((hs,), 0.1, 1.0)
        yield p + ".post_attention_layernorm.weight", t((hs,), 0.1, 1.0)
        yield from lin(p + ".self_attn.q_proj", dc.num_attention_heads * dc.head_dim, hs, False, 1.5)
        yield from lin(p + ".self_attn.k_proj", dc.num_key_value_heads * dc.head_dim, hs, False, 1.5)
        yield from lin(p + ".self_attn.v_proj", dc.num_key_value_heads * dc.head_dim, hs, False)
        yield from lin(p + ".self_attn.o_proj", hs, dc.num_attention_heads * dc.head_dim, False)
        yield from lin(p + ".mlp.gate_proj", dc.intermediate_size, hs, False)
        yield from lin(p + ".mlp.up_proj", dc.intermediate_size, hs, False)
        yield from lin(p + ".mlp.down_proj", hs, dc.intermediate_size, False)
        yield p + ".self_attn_layer_scale.scale", t((hs,), 0.2, 0.5)
        yield p + ".mlp_layer_scale.scale", t((hs,), 0.2, 0.5)
    ld = dc.latent_dim
    for i, f in enumerate(dc.upsampling_ratios):
        p = f"decoder.upsample.{i}.layers"
        yield from conv(p + ".0.conv", ld, f, ld, math.sqrt(f))
        yield from conv(p + ".1.dwconv.conv", ld, 7, 1)
        yield p + ".1.norm.weight", t((ld,), 0.1, 1.0)
        yield p + ".1.norm.bias", t((ld,), 0.05)
        yield from lin(p + ".1.pwconv1", 4 * ld, ld)
        yield from lin(p + ".1.pwconv2", ld, 4 * ld)
        yield p + ".1.gamma", t((ld,), 0.2, 0.4)
    yield from conv("decoder.decoder.0.conv", dc.decoder_dim, 7, ld)
    for bi, rate in enumerate(dc.upsample_rates):
        cin, cout = dc.decoder_dim >> bi, dc.decoder_dim >> (bi + 1)
        p = f"decoder.decoder.{bi + 1}.block"
        yield p + ".0.alpha", t((cin,), 0.5)
        yield p + ".0.beta", t((cin,), 0.5)
        yield from conv(p + ".1.conv", cout, 2 * rate, cin, math.sqrt(rate) * 0.8)
        for ri in range(3):
            q = f"{p}.{ri + 2}"
            yield q + ".act1.alpha", t((cout,), 0.5)
            yield q + ".act1.beta", t((cout,), 0.5)
            yield from conv(q + ".conv1.conv", cout, 7, cout, 0.7)
            yield q + ".act2.alpha", t((cout,), 0.5)
            yield q + ".act2.beta", t((cout,), 0.5)
            yield from conv(q + ".conv2.conv", cout, 1, cout, 0.3)
    n, cl = len(dc.upsample_rates), dc.decoder_dim >> len(dc.upsample_rates)
    yield f"decoder.decoder.{n + 1}.alpha", t((cl,), 0.5)
    yield f"decoder.decoder.{n + 1}.beta", t((cl,), 0.5)
    yield from conv(f"decoder.decoder.{n + 2}.conv", 1, 7, cl, 0.5)


def mlx_affine_quantize(w, group_size: int = 64, bits: int = 8):
    """A [N, K] float matrix in MLX's affine-quantised checkpoint form (mlx.core.quantize [3P]): uint32 words [N, K*bits/32]
    (element i of a row in word i // (32/bits) at bit bits * (i % (32/bits))), bf16 scales and biases [N, K/group_size] as torch
    tensors.  Host utility for the synthetic quantised benches (there are no checkpoints offline); the arithmetic follows mlx:
    scale = (max - min) / (2^bits - 1) signed so that the group's larger-magnitude extreme is a code point."""
    import torch
    w = np.asarray(w, np.float32)
    N, K = w.shape
    n_bins = float(2 ** bits - 1)
    g = w.reshape(N, K // group_size, group_size)
    w_max, w_min = g.max(-1), g.min(-1)
    mask = np.abs(w_min) > np.abs(w_max)
    scales = np.maximum((w_max - w_min) / n_bins, 1e-7).astype(np.float32)
    scales = np.where(mask, scales, -scales)
    edge = np.where(mask, w_min, w_max)
    q0 = np.round(edge / scales)
    scales = np.where(q0 != 0, edge / np.where(q0 != 0, q0, 1), scales).astype(np.float32)
    biases = np.where(q0 == 0, 0.0, edge).astype(np.float32)
    q = np.clip(np.round((g - biases[..., None]) / scales[..., None]), 0, n_bins).astype(np.uint32).reshape(N, K)
    epw = 32 // bits
    words = np.zeros((N, K // epw), np.uint32)
    for j in range(epw):
        words |= q[:, j::epw] << np.uint32(bits * j)
    return words, torch.from_numpy(scales).bfloat16(), torch.from_numpy(biases).bfloat16()


def qwen3tts_reference_synthetic_weights(cfg, seed: int = 717):
    """Synthetic checkpoint tensors of the in-context voice-cloning front end in the engine's key layout (`speaker_encoder.*` as
    Qwen3TTSSpeakerEncoder.sanitize leaves them, `encoder_model.*` as Qwen3TTSSpeechTokenizer.sanitize does): for benches and tools.
    cfg: Qwen3TTSConfiguration with speaker_encoder / tokenizer_encoder set.  Yields (name, float32 array)."""
    key = [seed * 100000]

    def t(shape, amp):
        key[0] += 1
        return synth_tensor(key[0], shape, amp)

    def conv(p, co, k, ci, bias=True, gain=1.0):
        yield p + ".weight", t((co, k, ci), gain * math.sqrt(3.0 / (k * ci)))
        if bias:
            yield p + ".bias", t((co,), 0.05)
    sp = cfg.speaker_encoder
    if sp is not None:
        ch, ks = sp.enc_channels, sp.enc_kernel_sizes
        P = "speaker_encoder."
        yield from conv(P + "blocks.0.conv", ch[0], ks[0], sp.mel_dim, gain=1.5)
        for i in range(1, len(ch) - 1):
            p = f"{P}blocks.{i}"
            w = ch[i] // sp.enc_res2net_scale
            yield from conv(p + ".tdnn1.conv", ch[i], 1, ch[i - 1], gain=1.5)
            for j in range(sp.enc_res2net_scale - 1):
                yield from conv(f"{p}.res2net_block.blocks.{j}.conv", w, ks[i], w, gain=1.5)
            yield from conv(p + ".tdnn2.conv", ch[i], 1, ch[i], gain=1.5)
            yield from conv(p + ".se_block.conv1", sp.enc_se_channels, 1, ch[i])
            yield from conv(p + ".se_block.conv2", ch[i], 1, sp.enc_se_channels)
        yield from conv(P + "mfa.conv", ch[-1], ks[-1], ch[-1], gain=1.5)
        yield from conv(P + "asp.tdnn.conv", sp.enc_attention_channels, 1, 3 * ch[-1])
        yield from conv(P + "asp.conv", ch[-1], 1, sp.enc_attention_channels, gain=2.0)
        yield from conv(P + "fc", sp.enc_dim, 1, 2 * ch[-1])
    en = cfg.tokenizer_encoder
    if en is not None:
        P = "encoder_model."
        nf, mult = en.num_filters, 1
        yield from conv(P + "encoder.init_conv1d.conv.conv", nf, en.kernel_size, en.audio_channels, gain=2.0)
        for li, ratio in enumerate(reversed(en.upsampling_ratios)):
            p = f"{P}encoder.layers.{li}"
            dim = mult * nf
            for ri in range(en.num_residual_layers):
                yield from conv(f"{p}.residuals.{ri}.block.0.conv.conv", dim // en.compress, en.residual_kernel_size, dim, gain=1.3)
                yield from conv(f"{p}.residuals.{ri}.block.1.conv.conv", dim, 1, dim // en.compress, gain=0.7)
                if en.use_conv_shortcut:
                    yield from conv(f"{p}.residuals.{ri}.shortcut.conv.conv", dim, 1, dim)
            yield from conv(p + ".downsample.conv.conv", 2 * dim, 2 * ratio, dim, gain=1.3)
            mult *= 2
        D, I = en.hidden_size, en.intermediate_size
        yield from conv(P + "encoder.final_conv1d.conv.conv", D, en.last_kernel_size, mult * nf, gain=1.3)
        for li in range(en.num_hidden_layers):
            p = f"{P}encoder_transformer.transformer.layers.{li}"
            for nm in ("norm1", "norm2"):
                yield f"{p}.{nm}.weight", (1.0 + t((D,), 0.2)).astype(np.float32)
                yield f"{p}.{nm}.bias", t((D,), 0.1)
            yield p + ".self_attn.in_proj.weight", t((3 * D, D), math.sqrt(3.0 / D))
            yield p + ".self_attn.out_proj.weight", t((D, D), math.sqrt(3.0 / D))
            yield p + ".gating.linear1.weight", t((I, D), math.sqrt(3.0 / D))
            yield p + ".gating.linear2.weight", t((D, I), math.sqrt(3.0 / I))
            yield p + ".layer_scale_1.scale", (0.3 + t((D,), 0.1)).astype(np.float32)
            yield p + ".layer_scale_2.scale", (0.3 + t((D,), 0.1)).astype(np.float32)
        ds = max(1, int((en.sampling_rate / float(np.prod(en.upsampling_ratios))) / en.frame_rate))
        yield from conv(P + "downsample.conv.conv.conv", D, 2 * ds, D, bias=False)
        keep = min(cfg.encoder_valid_num_quantizers, en.num_quantizers)
        for grp, nq in (("rvq_first", 1), ("rvq_rest", max(keep - 1, 0))):
            p = f"{P}quantizer.{grp}"
            if nq == 0:
                continue
            yield p + ".input_proj.weight", t((en.codebook_dim, 1, D), math.sqrt(3.0 / D))
            for i in range(nq):
                q = f"{p}.vq.layers.{i}.codebook"
                usage = (1.0 + np.abs(t((en.codebook_size,), 1.0))).astype(np.float32)
                yield q + ".cluster_usage", usage
                yield q + ".embedding_sum", (t((en.codebook_size, en.codebook_dim), 1.0 / (i + 1)) * usage[:, None]).astype(np.float32)
