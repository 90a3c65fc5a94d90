"""Pin the Llama oracle against HF transformers' LlamaForCausalLM (independent implementation
of the same architecture incl. llama3 rope scaling) on CPU, float32."""
import numpy as np
import pytest
import torch

from oracle import llama


def _hf_model(cfg, W):
    from transformers import LlamaConfig as HFConfig, LlamaForCausalLM
    hf = HFConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                  intermediate_size=cfg.intermediate_size, num_attention_heads=cfg.num_attention_heads,
                  num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.resolved_head_dim,
                  rms_norm_eps=cfg.rms_norm_eps, vocab_size=cfg.vocab_size, rope_theta=cfg.rope_theta,
                  rope_scaling=dict(cfg.rope_scaling), tie_word_embeddings=cfg.tie_word_embeddings,
                  max_position_embeddings=cfg.max_position_embeddings, attention_bias=False, mlp_bias=False,
                  attn_implementation="eager")
    m = LlamaForCausalLM(hf).to(torch.float32).eval()
    sd = {k: v.to(torch.float32) for k, v in W.items()}
    if cfg.tie_word_embeddings:
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "rotary" not in k], missing
    assert not unexpected, unexpected
    return m


def test_llama3_freqs_match_hf_inv_freq():
    cfg = llama.ORPHEUS_3B
    from transformers import LlamaConfig as HFConfig
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    hf = HFConfig(hidden_size=cfg.hidden_size, num_attention_heads=cfg.num_attention_heads, head_dim=128,
                  rope_theta=cfg.rope_theta, rope_scaling=dict(cfg.rope_scaling),
                  max_position_embeddings=cfg.max_position_embeddings)
    inv, scale = ROPE_INIT_FUNCTIONS["llama3"](hf, "cpu")
    assert scale == 1.0
    ours = 1.0 / llama.llama3_freqs(cfg)
    np.testing.assert_allclose(ours, inv.numpy(), rtol=2e-6)
    # scaling really is active for the 3B config: lowest frequencies divided by 32
    plain = cfg.rope_theta ** (-np.arange(0, 128, 2) / 128.0)
    assert np.isclose(ours[-1], plain[-1] / 32.0, rtol=1e-5)
    assert np.isclose(ours[0], plain[0], rtol=1e-6)


@pytest.mark.parametrize("tied", [True, False])
def test_prefill_and_decode_match_hf_fp32(tied):
    cfg = llama.LlamaConfig(**{**llama.TINY.__dict__, "tie_word_embeddings": tied})
    W = llama.make_synthetic_weights(cfg, dtype=torch.float32)
    hf = _hf_model(cfg, W)
    rng = np.random.default_rng(3)
    ids = rng.integers(0, cfg.vocab_size, (2, 9))
    with torch.no_grad():
        ref = hf(torch.from_numpy(ids)).logits.numpy()               # [2, 9, V]
    o = llama.LlamaOracle(cfg, W, round=None)
    o.reset(2)
    # prefill 5 tokens at once, then 4 single-token decode steps through the KV cache
    got = [o.forward([ids[0, :5], ids[1, :5]])]
    for t in range(5, 9):
        got.append(o.forward([ids[0, t:t + 1], ids[1, t:t + 1]]))
    for r in range(2):
        full = np.concatenate([g[r].numpy() for g in got], 0)
        np.testing.assert_allclose(full, ref[r], rtol=2e-4, atol=2e-4)
    assert np.abs(ref).max() > 1.0                                    # logits are not degenerate


def test_bf16_mode_stays_close_to_fp32_and_is_bf16_valued():
    cfg = llama.TINY
    W = llama.make_synthetic_weights(cfg)                             # bf16 weights
    rng = np.random.default_rng(4)
    ids = [rng.integers(0, cfg.vocab_size, 6), rng.integers(0, cfg.vocab_size, 3)]
    a = llama.LlamaOracle(cfg, W, round="bf16"); a.reset(2)
    b = llama.LlamaOracle(cfg, W, round=None); b.reset(2)
    la, lb = a.forward(ids), b.forward(ids)
    for x, y in zip(la, lb):
        assert torch.equal(x, x.to(torch.bfloat16).to(torch.float32))
        err = (x - y).abs().max().item() / y.abs().max().item()
        assert err < 0.05, err
    # rows are independent: ragged lengths, separate offsets
    assert a.offset == [6, 3]


def test_qwen3_style_variant_matches_hf_qwen3():
    """q/k per-head RMSNorm + plain RoPE (Soprano / VyvoTTS LM) vs HF Qwen3ForCausalLM, float32."""
    from transformers import Qwen3Config, Qwen3ForCausalLM
    cfg = llama.TINY_QWEN3
    W = llama.make_synthetic_weights(cfg, dtype=torch.float32)
    hc = Qwen3Config(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                     intermediate_size=cfg.intermediate_size, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, rms_norm_eps=cfg.rms_norm_eps,
                     vocab_size=cfg.vocab_size, rope_theta=cfg.rope_theta, tie_word_embeddings=False,
                     max_position_embeddings=4096, attention_bias=False, attn_implementation="eager",
                     use_sliding_window=False)
    m = Qwen3ForCausalLM(hc).to(torch.float32).eval()
    missing, unexpected = m.load_state_dict({k: v.to(torch.float32) for k, v in W.items()}, strict=False)
    assert not unexpected and not [k for k in missing if "rotary" not in k], (missing, unexpected)
    rng = np.random.default_rng(5)
    ids = rng.integers(0, cfg.vocab_size, (1, 11))
    with torch.no_grad():
        out = m(torch.from_numpy(ids), output_hidden_states=True)
        ref = out.logits.numpy()[0]
    o = llama.LlamaOracle(cfg, W, round=None)
    o.reset(1)
    a = o.forward([ids[0, :6]])[0].numpy()
    b = np.concatenate([o.forward([ids[0, t:t + 1]])[0].numpy() for t in range(6, 11)])
    np.testing.assert_allclose(np.concatenate([a, b]), ref, rtol=3e-4, atol=3e-4)
    np.testing.assert_allclose(o.last_hidden.numpy()[-1], out.hidden_states[-1].numpy()[0, -1], rtol=3e-4, atol=3e-4)


def test_orpheus_3b_width_matches_hf_fp32():
    """The same pin at the BENCHMARKED per-layer dimensions (hidden 3072, 24 / 8 heads x 128, ffn 8192, llama3 rope scaling, tied head;
    one layer, vocabulary cut to 20 000 rows so that the CPU suite stays quick): prefill + decode through the cache against HF's
    LlamaForCausalLM, float32."""
    cfg = llama.LlamaConfig(num_hidden_layers=1, vocab_size=20000)         # every other field = ORPHEUS_3B
    assert (cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.resolved_head_dim) == (3072, 8192, 24, 8, 128)
    W = llama.make_synthetic_weights(cfg, dtype=torch.float32)
    hf = _hf_model(cfg, W)
    rng = np.random.default_rng(4)
    ids = rng.integers(0, cfg.vocab_size, (1, 7))
    with torch.no_grad():
        ref = hf(torch.from_numpy(ids)).logits.numpy()
    o = llama.LlamaOracle(cfg, W, round=None)
    o.reset(1)
    got = [o.forward([ids[0, :4]])] + [o.forward([ids[0, t:t + 1]]) for t in range(4, 7)]
    full = np.concatenate([g[0].numpy() for g in got], 0)
    np.testing.assert_allclose(full, ref[0], rtol=3e-4, atol=3e-4)
    assert np.abs(ref).max() > 1.0
