"""CPU: pin for oracle/mimi_encoder.py (the Qwen3-TTS speech-tokenizer encoder, not yet built on the device): HF transformers'
MimiModel.encode with the oracle's synthetic weights (SEANet encoder, transformer, downsampling conv, split residual VQ)."""
import numpy as np
import torch

from oracle import mimi_encoder as om


def _hf_model(cfg, W):
    from transformers import MimiConfig, MimiModel
    hd = cfg.hidden_size // cfg.num_attention_heads
    hc = MimiConfig(sampling_rate=cfg.sampling_rate, frame_rate=cfg.frame_rate, audio_channels=1, hidden_size=cfg.hidden_size,
                    num_filters=cfg.num_filters, num_residual_layers=cfg.num_residual_layers, upsampling_ratios=list(cfg.upsampling_ratios),
                    kernel_size=cfg.kernel_size, last_kernel_size=cfg.last_kernel_size, residual_kernel_size=cfg.residual_kernel_size,
                    dilation_growth_rate=cfg.dilation_growth_rate, use_causal_conv=True, pad_mode="constant", compress=cfg.compress,
                    codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim, num_quantizers=cfg.num_quantizers, use_conv_shortcut=False,
                    vector_quantization_hidden_dimension=cfg.codebook_dim, num_semantic_quantizers=1, num_hidden_layers=cfg.num_hidden_layers,
                    intermediate_size=cfg.intermediate_size, num_attention_heads=cfg.num_attention_heads,
                    num_key_value_heads=cfg.num_attention_heads, head_dim=hd, hidden_act="gelu", norm_eps=1e-5, rope_theta=cfg.rope_theta,
                    sliding_window=cfg.sliding_window, layer_scale_initial_scale=0.01, attention_bias=False, upsample_groups=cfg.hidden_size)
    hf = MimiModel(hc).eval()
    sd = hf.state_dict()
    t = lambda k: torch.from_numpy(np.asarray(W[k]))

    def put_conv(hf_name, o_name, bias=True):
        sd[hf_name + ".conv.weight"] = t(o_name + ".conv.conv.weight").permute(0, 2, 1).contiguous()
        if bias:
            sd[hf_name + ".conv.bias"] = t(o_name + ".conv.conv.bias")
    # HF flattens the SEANet encoder into one Sequential: conv, then per ratio [resnet block, ELU, strided conv], ELU, conv
    put_conv("encoder.layers.0", "encoder.init_conv1d")
    idx = 1
    for li in range(len(cfg.upsampling_ratios)):
        for ri in range(cfg.num_residual_layers):
            put_conv(f"encoder.layers.{idx}.block.1", f"encoder.layers.{li}.residuals.{ri}.block.0")
            put_conv(f"encoder.layers.{idx}.block.3", f"encoder.layers.{li}.residuals.{ri}.block.1")
            idx += 1
        idx += 1                                                   # ELU
        put_conv(f"encoder.layers.{idx}", f"encoder.layers.{li}.downsample")
        idx += 1
    idx += 1                                                       # ELU
    put_conv(f"encoder.layers.{idx}", "encoder.final_conv1d")
    D, H = cfg.hidden_size, cfg.num_attention_heads
    perm = torch.cat([torch.arange(0, hd, 2), torch.arange(1, hd, 2)])   # interleaved (MLX traditional RoPE) -> half layout (HF rotate_half)
    rows = torch.cat([h * hd + perm for h in range(H)])
    for li in range(cfg.num_hidden_layers):
        p, q = f"encoder_transformer.transformer.layers.{li}", f"encoder_transformer.layers.{li}"
        w = t(p + ".self_attn.in_proj.weight")
        sd[q + ".self_attn.q_proj.weight"] = w[:D][rows]
        sd[q + ".self_attn.k_proj.weight"] = w[D:2 * D][rows]
        sd[q + ".self_attn.v_proj.weight"] = w[2 * D:]
        sd[q + ".self_attn.o_proj.weight"] = t(p + ".self_attn.out_proj.weight")
        sd[q + ".mlp.fc1.weight"] = t(p + ".gating.linear1.weight")
        sd[q + ".mlp.fc2.weight"] = t(p + ".gating.linear2.weight")
        sd[q + ".input_layernorm.weight"], sd[q + ".input_layernorm.bias"] = t(p + ".norm1.weight"), t(p + ".norm1.bias")
        sd[q + ".post_attention_layernorm.weight"], sd[q + ".post_attention_layernorm.bias"] = t(p + ".norm2.weight"), t(p + ".norm2.bias")
        sd[q + ".self_attn_layer_scale.scale"], sd[q + ".mlp_layer_scale.scale"] = t(p + ".layer_scale_1.scale"), t(p + ".layer_scale_2.scale")
    sd["downsample.conv.weight"] = t("downsample.conv.conv.conv.weight").permute(0, 2, 1).contiguous()
    for hf_grp, o_grp, nq in (("semantic", "rvq_first", 1), ("acoustic", "rvq_rest", cfg.num_quantizers - 1)):
        hp, op = f"quantizer.{hf_grp}_residual_vector_quantizer", f"quantizer.{o_grp}"
        sd[hp + ".input_proj.weight"] = t(op + ".input_proj.weight").permute(0, 2, 1).contiguous()
        for i in range(nq):
            sd[f"{hp}.layers.{i}.codebook.embed_sum"] = t(f"{op}.vq.layers.{i}.codebook.embedding_sum")
            sd[f"{hp}.layers.{i}.codebook.cluster_usage"] = t(f"{op}.vq.layers.{i}.codebook.cluster_usage")
            sd[f"{hp}.layers.{i}.codebook.initialized"] = torch.ones(1)
    hf.load_state_dict(sd)
    return hf


def test_encode_matches_hf_mimi():
    cfg = om.TINY
    W = om.make_synthetic_weights(cfg)
    o = om.MimiEncoderOracle(cfg, W)
    hf = _hf_model(cfg, W)
    audio = (0.3 * np.random.default_rng(1).standard_normal((2, 1, 6 * 2 * 23))).astype(np.float32)    # 23 frames after downsampling
    codes, hidden = o.encode(audio, return_hidden=True)
    with torch.no_grad():
        x = hf.encoder(torch.from_numpy(audio))
        x = hf.encoder_transformer(x.transpose(1, 2))[0].transpose(1, 2)
        x = hf.downsample(x)
        hf_codes = hf.quantizer.encode(x).transpose(0, 1).numpy()                                     # [nq, B, T] -> [B, nq, T]
    assert hidden.shape == tuple(x.shape) == (2, cfg.hidden_size, 23)
    np.testing.assert_allclose(hidden, x.numpy(), rtol=2e-4, atol=2e-5)
    assert codes.shape == (2, cfg.valid_num_quantizers, 23)
    agree = (codes == hf_codes[:, : cfg.valid_num_quantizers]).mean()
    assert agree > 0.97, agree                                   # a float-level near-tie in one group shifts the residual of the later ones


def test_lengths_and_the_kept_code_groups():
    cfg = om.TINY
    o = om.MimiEncoderOracle(cfg, om.make_synthetic_weights(cfg))
    for n, frames in ((12, 1), (13, 2), (100, 9), (240, 20)):          # hop 6, then stride 2, both rounded up by the extra right padding
        c = o.encode(np.zeros((1, 1, n), np.float32))
        assert c.shape == (1, cfg.valid_num_quantizers, frames), (n, c.shape)
