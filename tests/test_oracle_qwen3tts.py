"""CPU: pins for oracle/qwen3tts.py (a22).  The reference cannot run here ("parity unpinned"); these tests pin the
restatement against independent formulations and the structural facts the reference relies on."""
import numpy as np
import torch

from oracle import qwen3tts as oq


def test_causal_transposed_conv_matches_naive_scatter():
    # ConvTransposed1d(padding 0) + right trim (Qwen3TTSSpeechTokenizer.swift:533-551): y[n*s + j] += x[c, n] * w[o, j, c]
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 3, 5)).astype(np.float32)
    for s, k in ((2, 4), (3, 6), (2, 2)):
        w = rng.standard_normal((4, k, 3)).astype(np.float32)
        b = rng.standard_normal(4).astype(np.float32)
        ref = np.zeros((4, (5 - 1) * s + k), np.float64)
        for n in range(5):
            for j in range(k):
                ref[:, n * s + j] += w[:, j, :].astype(np.float64) @ x[0, :, n]
        ref = (ref + b[:, None])[:, : 5 * s]
        got = oq.causal_conv_transpose1d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), s).numpy()[0]
        assert got.shape == (4, 5 * s) and np.abs(got - ref).max() < 1e-5


def test_decoder_is_causal_so_streaming_equals_full_decode():
    cfg = oq.TINY.decoder
    dec = oq.SpeechDecoderOracle(cfg, oq.make_synthetic_decoder_weights(cfg))
    rng = np.random.default_rng(1)
    codes = rng.integers(0, cfg.codebook_size, (2, cfg.num_quantizers, 9))
    full = dec.decode(codes)
    assert full.shape == (2, 9 * cfg.total_upsample) and np.abs(full).max() <= 1.0 and np.abs(full).std() > 1e-3
    for n in (1, 4):                              # a prefix of the frames gives exactly the prefix of the waveform
        part = dec.decode(codes[:, :, :n])
        assert np.abs(part - full[:, : n * cfg.total_upsample]).max() < 2e-5
    changed = codes.copy(); changed[:, :, 6] = (changed[:, :, 6] + 1) % cfg.codebook_size
    alt = dec.decode(changed)
    assert np.array_equal(alt[:, : 6 * cfg.total_upsample], full[:, : 6 * cfg.total_upsample]) or \
        np.abs(alt[:, : 6 * cfg.total_upsample] - full[:, : 6 * cfg.total_upsample]).max() < 2e-5
    assert np.abs(alt - full).max() > 1e-4


def _stream(dec, codes, cuts):
    dec.reset_streaming_state()
    out, a = [], 0
    for b in list(cuts) + [codes.shape[-1]]:
        if b > a:
            out.append(dec.streaming_step(codes[:, :, a:b]))
        a = b
    return np.concatenate(out, -1)


def test_streaming_step_restates_the_reference_step_functions():
    """streamingStep (Qwen3TTSSpeechTokenizer.swift:971-1006) with carried conv buffers, transposed-conv overflow and KV cache.
    (1) One chunk is the whole-sequence decode.  (2) With the transposed-conv biases of the decoder blocks zeroed, ANY chunking is
    the whole-sequence decode (carried state is exact).  (3) With the biases, the reference's overlap-add sums two biased
    outputs (:556-559): the difference to the whole-sequence decode starts exactly at the first chunk boundary."""
    cfg = oq.TINY.decoder
    W = oq.make_synthetic_decoder_weights(cfg)
    rng = np.random.default_rng(5)
    codes = rng.integers(0, cfg.codebook_size, (2, cfg.num_quantizers, 11))
    up = cfg.total_upsample
    dec = oq.SpeechDecoderOracle(cfg, W)
    full = dec.decode(codes)
    assert np.abs(_stream(dec, codes, []) - full).max() < 2e-5
    W0 = dict(W)
    for bi in range(len(cfg.upsample_rates)):
        W0[f"decoder.decoder.{bi + 1}.block.1.conv.bias"] = np.zeros_like(W[f"decoder.decoder.{bi + 1}.block.1.conv.bias"])
    dec0 = oq.SpeechDecoderOracle(cfg, W0)
    full0 = dec0.decode(codes)
    for cuts in ([4], [1, 2, 3], [5, 6, 10], list(range(1, 11))):
        assert np.abs(_stream(dec0, codes, cuts) - full0).max() < 3e-5, cuts
    got = _stream(dec, codes, [4, 9])
    assert got.shape == full.shape
    assert np.abs(got[:, : 4 * up] - full[:, : 4 * up]).max() < 2e-5          # nothing differs before the first boundary
    assert np.abs(got[:, 4 * up: 4 * up + up] - full[:, 4 * up: 4 * up + up]).max() > 1e-4     # the doubled bias right after it


def test_sample_token_set_semantics():
    rng = np.random.default_rng(2)
    V = 300
    lg = torch.as_tensor((rng.standard_normal(V) * 3).astype(np.float32)).bfloat16().float().numpy()
    top = np.argsort(lg)[::-1]
    sup = list(range(V - 50, V))
    # greedy honours suppress + penalty
    lg2 = lg.copy(); lg2[V - 1] = 50.0
    assert oq.sample_token(lg2, 0.0, 1.0, 0, 1.0, None, sup, None, 0.0, 1, 0, 0) == int(np.argmax(np.where(np.arange(V) < V - 50, lg2, -np.inf)))
    big = lg.copy(); big[top[0]] = abs(big[top[0]]) + 1
    assert oq.sample_token(big, 0.0, 1.0, 0, 100.0, [int(top[0])], None, None, 0.0, 1, 0, 0) != int(top[0])
    # top-k: samples only from the k best; EOS restored even when outside the top-k
    seen = {oq.sample_token(lg, 1.5, 1.0, 5, 1.0, None, None, None, 0.0, 7, 0, s) for s in range(300)}
    assert seen <= set(int(t) for t in top[:5]) and len(seen) >= 3
    eos = int(top[100])
    hot = lg.copy(); hot[eos] = lg[top[0]] + 2
    seen = {oq.sample_token(hot, 1.0, 1.0, 5, 1.0, None, None, eos, 0.0, 7, 0, s) for s in range(100)}
    assert eos in seen
    # min-p removes everything far below the maximum
    seen = {oq.sample_token(lg, 1.0, 1.0, 0, 1.0, None, None, None, 0.5, 7, 0, s) for s in range(200)}
    assert all(lg[t] >= lg.max() + np.log(0.5) - 0.05 for t in seen)
    # top-p keeps the head of the distribution
    seen = {oq.sample_token(lg, 1.0, 0.5, 0, 1.0, None, None, None, 0.0, 7, 0, s) for s in range(300)}
    p = np.exp(lg - lg.max()); p /= p.sum()
    assert p[list(seen)].min() >= np.sort(p)[::-1][np.searchsorted(np.cumsum(np.sort(p)[::-1]), 0.5) + 1] * 0.9


def test_frame_loop_is_deterministic_and_teacher_forcing_reproduces_logits():
    for cfg in (oq.TINY, oq.TINY_PROJ):
        W = oq.make_synthetic_weights(cfg)
        m = oq.Qwen3TTSOracle(cfg, W)
        text_ids = [5, 6, 7, cfg.tts_pad_token_id, cfg.tts_pad_token_id, cfg.tts_bos_token_id, 9]
        codec_ids = [-1, -1, -1, cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id, cfg.codec_bos_id]
        params = dict(temperature=0.9, top_p=1.0, top_k=50, repetition_penalty=1.05, min_p=0.0, seed=3)
        a, la = m.generate_row(text_ids, codec_ids, [11, 12, cfg.tts_eos_token_id], params, max_frames=5)
        b, lb = m.generate_row(text_ids, codec_ids, [11, 12, cfg.tts_eos_token_id], params, max_frames=5)
        assert a.shape == (5, cfg.num_code_groups) and np.array_equal(a, b)
        assert a[:, 0].max() < cfg.talker.vocab_size - 1024 and a[:, 1:].max() < cfg.predictor.vocab_size
        c, lc = m.generate_row(text_ids, codec_ids, [11, 12, cfg.tts_eos_token_id], params, max_frames=5, forced_codes=a)
        assert np.array_equal(c, a) and all(np.array_equal(x, y) for x, y in zip(la, lc[:5]))


def test_decoder_building_blocks_match_hf_code2wav_modules():
    """Independent implementation of the shared building blocks: HF transformers' Qwen3-Omni Code2Wav (the same vocoder family:
    SnakeBeta, causal conv, ConvNeXt block, residual unit).  NOT compared: the transposed conv - HF trims kernel - stride samples
    on BOTH sides, the reference on the right only (Qwen3TTSSpeechTokenizer.swift:533-551,732-749), which is what the oracle
    restates and test_causal_transposed_conv_matches_naive_scatter pins."""
    from transformers.models.qwen3_omni_moe import modeling_qwen3_omni_moe as hf
    rng = np.random.default_rng(5)
    C, T = 12, 37
    x = torch.from_numpy(rng.standard_normal((2, C, T)).astype(np.float32))

    def r(*shape, s=0.3):
        return torch.from_numpy((rng.standard_normal(shape) * s).astype(np.float32))

    with torch.no_grad():
        # SnakeBeta
        sb = hf.Qwen3OmniMoeSnakeBeta(C)
        sb.alpha.copy_(r(C)); sb.beta.copy_(r(C))
        assert torch.allclose(oq.snake_beta(x, sb.alpha, sb.beta), sb(x), atol=1e-6)
        # causal (dilated) conv
        for k, dil in ((7, 1), (7, 3), (7, 9), (1, 1)):
            cc = hf.Qwen3OmniMoeCausalConvNet(C, C, k, dilation=dil)
            w = cc.conv.weight.permute(0, 2, 1).contiguous()                                   # oracle layout [Co, k, Ci]
            assert torch.allclose(oq.causal_conv1d(x, w, cc.conv.bias, dilation=dil), cc(x), atol=1e-5)
        # ConvNeXt block
        cn = hf.Qwen3OmniMoeConvNeXtBlock(C)
        cn.gamma.copy_(r(C, s=1.0)); cn.norm.weight.copy_(1.0 + r(C)); cn.norm.bias.copy_(r(C))
        W = {"p.dwconv.conv.weight": cn.dwconv.conv.weight.permute(0, 2, 1).contiguous(), "p.dwconv.conv.bias": cn.dwconv.conv.bias,
             "p.norm.weight": cn.norm.weight, "p.norm.bias": cn.norm.bias, "p.pwconv1.weight": cn.pwconv1.weight,
             "p.pwconv1.bias": cn.pwconv1.bias, "p.pwconv2.weight": cn.pwconv2.weight, "p.pwconv2.bias": cn.pwconv2.bias,
             "p.gamma": cn.gamma}
        o = oq.SpeechDecoderOracle(oq.DecoderConfig(), {k: v.detach().numpy() for k, v in W.items()})
        assert torch.allclose(o.convnext(x, "p"), cn(x), atol=1e-5)
        # residual unit (act1 -> conv k7 dilated -> act2 -> conv k1, + x), as inlined in SpeechDecoderOracle.decode
        ru = hf.Qwen3OmniMoeCode2WavDecoderResidualUnit(C, dilation=3)
        for a in (ru.act1, ru.act2):
            a.alpha.copy_(r(C)); a.beta.copy_(r(C))
        t = oq.snake_beta(x, ru.act1.alpha, ru.act1.beta)
        t = oq.causal_conv1d(t, ru.conv1.conv.weight.permute(0, 2, 1).contiguous(), ru.conv1.conv.bias, dilation=3)
        t = oq.snake_beta(t, ru.act2.alpha, ru.act2.beta)
        y = x + oq.causal_conv1d(t, ru.conv2.conv.weight.permute(0, 2, 1).contiguous(), ru.conv2.conv.bias)
        assert torch.allclose(y, ru(x), atol=1e-5)


def test_decoder_transformer_matches_hf_code2wav_transformer_inside_the_window():
    """The speech-tokenizer's pre-transformer (RMSNorm, RoPE, causal attention, layer scale, SwiGLU) against HF transformers'
    Qwen3-Omni Code2Wav transformer with the same weights.  HF masks with a 72-frame sliding window, the reference with a plain
    causal mask (Qwen3TTSSpeechTokenizer.swift:449-503: createAdditiveCausalMask), so the comparison uses T < 72; the oracle's
    input / output projections (which HF's module does not have) are set to the identity."""
    from transformers.models.qwen3_omni_moe import configuration_qwen3_omni_moe as hc, modeling_qwen3_omni_moe as hf
    D, H, hd, ff, L, T = 32, 4, 8, 48, 2, 40
    cfg = hc.Qwen3OmniMoeCode2WavConfig(hidden_size=D, num_attention_heads=H, num_key_value_heads=H, intermediate_size=ff,
                                        num_hidden_layers=L, sliding_window=72, rms_norm_eps=1e-5, layer_scale_initial_scale=0.01,
                                        max_position_embeddings=128, rope_parameters={"rope_type": "default", "rope_theta": 10000.0})
    cfg._attn_implementation = "eager"
    tm = hf.Qwen3OmniMoeCode2WavTransformerModel(cfg).eval()
    rng = np.random.default_rng(8)
    with torch.no_grad():
        for prm in tm.parameters():
            prm.copy_(torch.from_numpy((rng.standard_normal(tuple(prm.shape)) * (0.3 if prm.ndim > 1 else 1.0)).astype(np.float32))
                      + (1.0 if prm.ndim == 1 else 0.0))
    P = "decoder.pre_transformer"
    W = {P + ".input_proj.weight": np.eye(D, dtype=np.float32), P + ".input_proj.bias": np.zeros(D, np.float32),
         P + ".output_proj.weight": np.eye(D, dtype=np.float32), P + ".output_proj.bias": np.zeros(D, np.float32),
         P + ".norm.weight": tm.norm.weight.detach().numpy()}
    for i, lyr in enumerate(tm.layers):
        p = f"{P}.layers.{i}"
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            W[f"{p}.self_attn.{nm}.weight"] = getattr(lyr.self_attn, nm).weight.detach().numpy()
        for nm in ("gate_proj", "up_proj", "down_proj"):
            W[f"{p}.mlp.{nm}.weight"] = getattr(lyr.mlp, nm).weight.detach().numpy()
        W[p + ".input_layernorm.weight"] = lyr.input_layernorm.weight.detach().numpy()
        W[p + ".post_attention_layernorm.weight"] = lyr.post_attention_layernorm.weight.detach().numpy()
        W[p + ".self_attn_layer_scale.scale"] = lyr.self_attn_layer_scale.scale.detach().numpy()
        W[p + ".mlp_layer_scale.scale"] = lyr.mlp_layer_scale.scale.detach().numpy()
    dc = oq.DecoderConfig(latent_dim=D, hidden_size=D, intermediate_size=ff, head_dim=hd, num_attention_heads=H, num_hidden_layers=L,
                          num_key_value_heads=H, rms_norm_eps=1e-5, rope_theta=10000.0)
    o = oq.SpeechDecoderOracle(dc, W)
    x = torch.from_numpy(rng.standard_normal((2, T, D)).astype(np.float32))
    with torch.no_grad():
        ref = tm(inputs_embeds=x).last_hidden_state
        got = o.transformer(x)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4), float((got - ref).abs().max())
