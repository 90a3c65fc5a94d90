"""Orpheus token LM (Llama-3 architecture): CPU restatement of the reference's forward pass.
Test infrastructure only.

Follows (Sources/MLXAudioTTS/Models/Llama/):
  * Llama3ScaledRoPE                LlamaTTS.swift:104-202   (freqs :126-156, apply :188-201)
  * LlamaTTSAttention               LlamaTTS.swift:206-267   (GQA, RoPE at cache.offset, SDPA)
  * LlamaTTSMLP                     LlamaTTS.swift:271-285   down(silu(gate(x)) * up(x))
  * LlamaTTSTransformerBlock        LlamaTTS.swift:289-311   pre-norm residual block
  * LlamaTTSModelInner              LlamaTTS.swift:315-346   embed -> blocks -> norm
  * LlamaTTSModel.callAsFunction    LlamaTTS.swift:557-567   lm_head or tied embedding
  * LlamaTTSConfiguration           LlamaTTSConfig.swift:15-138
Third-party semantics restated [3P: mlx-swift 0.31.4 / mlx-swift-lm 3.31.4, not vendored]:
  RMSNorm      fp32:  n = T(x * rsqrt(mean(x^2) + eps));  out = T(w * n)        (T = model dtype)
  Linear       T(x @ W^T) with fp32 accumulation (no bias in Orpheus)
  fast.RoPE    non-traditional (rotate-half): pair (i, i+d/2), angle = pos / freqs[i]
  SDPA         softmax(scale * q k^T + mask) v, fp32 scores/softmax/accumulate, T at the output;
               GQA: n_q/n_kv consecutive q heads share one kv head; causal mask when L > 1
  KVCacheSimple  append at offset; offset counts every token fed (LlamaTTS.swift:249-251)
  silu         x * sigmoid(x), each primitive rounded to T
`round` = "bf16" reproduces MLX's storage dtype at every primitive boundary above (the
reference checkpoint mlx-community/orpheus-3b-0.1-ft-bf16 is bf16); `round` = None keeps
everything in float32 and is what the HF `transformers` cross-check compares against
(tests/test_oracle_llama.py).

Batching: the reference only ever runs B = 1 (LlamaTTS.swift:683-688).  Rows here are
independent utterances with their own positions; left-pad tokens are never fed, so each
row's RoPE position starts at 0 at its first real token - identical to B = 1 row by row
(SURVEY App. D.1).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch


@dataclass
class LlamaConfig:
    hidden_size: int = 3072
    num_hidden_layers: int = 28
    intermediate_size: int = 8192
    num_attention_heads: int = 24
    num_key_value_heads: int = 8
    head_dim: int | None = 128
    rms_norm_eps: float = 1e-5
    vocab_size: int = 156940
    rope_theta: float = 500000.0
    rope_traditional: bool = False
    rope_scaling: dict | None = field(default_factory=lambda: dict(
        factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
        original_max_position_embeddings=8192, rope_type="llama3"))
    tie_word_embeddings: bool = True
    max_position_embeddings: int = 131072
    # Qwen3-style variants: per-head RMSNorm of q/k before RoPE and plain RoPE(base)
    # (SopranoAttention Soprano.swift:24-97; VyvoTTS Qwen3.swift:204-205)
    qk_norm: bool = False
    rope_plain: bool = False
    # Qwen3-TTS talker / code predictor: RoPE written out as array ops in the model dtype,
    # q*cos + rotate_half(q)*sin with cos/sin cast to T (Qwen3TTSTalker.swift:15-24,92-95)
    rope_ops_in_dtype: bool = False

    @property
    def resolved_head_dim(self) -> int:
        return self.head_dim or self.hidden_size // self.num_attention_heads

    def to_json_dict(self) -> dict:
        d = dict(hidden_size=self.hidden_size, num_hidden_layers=self.num_hidden_layers,
                 intermediate_size=self.intermediate_size, num_attention_heads=self.num_attention_heads,
                 num_key_value_heads=self.num_key_value_heads, rms_norm_eps=self.rms_norm_eps,
                 vocab_size=self.vocab_size, rope_theta=self.rope_theta,
                 rope_traditional=self.rope_traditional, tie_word_embeddings=self.tie_word_embeddings,
                 max_position_embeddings=self.max_position_embeddings, model_type="llama")
        if self.head_dim is not None:
            d["head_dim"] = self.head_dim
        if self.rope_scaling is not None:
            d["rope_scaling"] = dict(self.rope_scaling)
        return d


ORPHEUS_3B = LlamaConfig()
TINY = LlamaConfig(hidden_size=768, num_hidden_layers=2, intermediate_size=1024, num_attention_heads=6,
                   num_key_value_heads=2, head_dim=128, vocab_size=1000 + 7 * 64)
TINY_QWEN3 = LlamaConfig(hidden_size=512, num_hidden_layers=2, intermediate_size=768, num_attention_heads=4,
                         num_key_value_heads=2, head_dim=128, vocab_size=1200, rope_theta=10000.0, rope_scaling=None,
                         tie_word_embeddings=False, qk_norm=True, rope_plain=True, rms_norm_eps=1e-6)
TINY64 = LlamaConfig(hidden_size=256, num_hidden_layers=3, intermediate_size=512, num_attention_heads=4,
                     num_key_value_heads=4, head_dim=64, vocab_size=777)


def llama3_freqs(cfg: LlamaConfig) -> np.ndarray:
    """LlamaTTS.swift:126-156, float32 arithmetic.  Returns freqs[d/2] (angle = pos / freqs)."""
    dims = cfg.resolved_head_dim
    f32 = np.float32
    idx = np.arange(0, dims, 2, dtype=np.float32)
    freqs = np.power(f32(cfg.rope_theta), idx / f32(dims)).astype(np.float32)
    rs = cfg.rope_scaling
    if rs is None:
        # the reference always builds Llama3ScaledRoPE with defaults 32/1/4/8192 (:161-185)
        rs = {}
    factor = f32(rs.get("factor", 32.0))
    low = f32(rs.get("low_freq_factor", 1.0))
    high = f32(rs.get("high_freq_factor", 4.0))
    old = f32(rs.get("original_max_position_embeddings", 8192.0))
    if cfg.rope_plain:                       # MLXNN.RoPE(dimensions, traditional: false, base) [3P]
        return freqs
    wavelens = f32(2.0 * np.float32(np.pi)) * freqs
    low_wl = old / low
    high_wl = old / high
    freqs = np.where(wavelens > low_wl, freqs * factor, freqs).astype(np.float32)
    is_med = np.logical_and(wavelens > high_wl, wavelens < low_wl)
    smooth = ((old / wavelens - low) / (high - low)).astype(np.float32)
    denom = ((f32(1.0) - smooth) / factor + smooth).astype(np.float32)
    smooth_freqs = (freqs / denom).astype(np.float32)
    return np.where(is_med, smooth_freqs, freqs).astype(np.float32)


def _rounder(mode):
    if mode is None:
        return lambda t: t
    if mode == "bf16":
        return lambda t: t.to(torch.bfloat16).to(torch.float32)
    raise ValueError(mode)


class LlamaOracle:
    """Weights: dict name -> torch tensor (any float dtype), HF/MLX key layout:
    model.embed_tokens.weight, model.layers.N.{input_layernorm,post_attention_layernorm}.weight,
    model.layers.N.self_attn.{q,k,v,o}_proj.weight, model.layers.N.mlp.{gate,up,down}_proj.weight,
    model.norm.weight, [lm_head.weight]."""

    def __init__(self, cfg: LlamaConfig, weights: dict, round: str | None = "bf16"):
        self.cfg = cfg
        self.r = _rounder(round)
        self.w = {k: torch.as_tensor(v).to(torch.float32) for k, v in weights.items()}
        self.inv_freqs = torch.from_numpy((np.float32(1.0) / llama3_freqs(cfg)).astype(np.float32))
        self.reset(0)

    def reset(self, batch: int):
        L = self.cfg.num_hidden_layers
        self.k_cache = [[None] * L for _ in range(batch)]     # per row, per layer [Hkv, S, D]
        self.v_cache = [[None] * L for _ in range(batch)]
        self.offset = [0] * batch

    # -- primitives -----------------------------------------------------------
    def rmsnorm(self, x, w):
        x = x.to(torch.float32)
        n = self.r(x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + self.cfg.rms_norm_eps))
        return self.r(w * n)

    def linear(self, x, w):
        return self.r(x @ w.t())

    def rope(self, x, positions):
        """x [H, L, D], positions [L] int -> rotate-half with llama3 freqs."""
        D = x.shape[-1]
        ang = positions.to(torch.float32)[:, None] * self.inv_freqs[None, :]       # [L, D/2]
        c, s = torch.cos(ang), torch.sin(ang)
        x1, x2 = x[..., : D // 2], x[..., D // 2:]
        if self.cfg.rope_ops_in_dtype:       # every array op rounds to T: T(T(x*cos) + T(rotate_half(x)*sin))
            c, s = self.r(c), self.r(s)
            return self.r(torch.cat([self.r(x1 * c) + self.r(-x2 * s), self.r(x2 * c) + self.r(x1 * s)], dim=-1))
        return self.r(torch.cat([x1 * c - x2 * s, x1 * s + x2 * c], dim=-1))

    # -- one row, L new tokens --------------------------------------------------
    def _forward_row(self, row: int, ids: torch.Tensor, logit_positions=None):
        return self.forward_embeds(row, self.w["model.embed_tokens.weight"][ids], logit_positions=logit_positions)   # [L, d]

    def forward_embeds(self, row: int, h: torch.Tensor, head: torch.Tensor | None = None, logit_positions=None):
        """L new positions given as input embeddings [L, d] (Qwen3TTSTalkerModel.callAsFunction takes
        inputsEmbeds, Qwen3TTSTalker.swift:274-305).  `head` overrides the output projection.
        logit_positions (test economy at full vocabulary width): indices among the L new positions whose logits are
        wanted - the output projection is a per-position Linear, so skipping positions changes no value."""
        cfg = self.cfg
        H, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.resolved_head_dim
        h = torch.as_tensor(h, dtype=torch.float32)
        L = h.shape[0]
        off = self.offset[row]
        pos = torch.arange(off, off + L)
        scale = float(D) ** -0.5
        for li in range(cfg.num_hidden_layers):
            p = f"model.layers.{li}"
            x = self.rmsnorm(h, self.w[p + ".input_layernorm.weight"])
            q = self.linear(x, self.w[p + ".self_attn.q_proj.weight"]).view(L, H, D).transpose(0, 1)
            k = self.linear(x, self.w[p + ".self_attn.k_proj.weight"]).view(L, Hkv, D).transpose(0, 1)
            v = self.linear(x, self.w[p + ".self_attn.v_proj.weight"]).view(L, Hkv, D).transpose(0, 1)
            if cfg.qk_norm:                  # Soprano.swift:75-76
                q = self.rmsnorm(q, self.w[p + ".self_attn.q_norm.weight"])
                k = self.rmsnorm(k, self.w[p + ".self_attn.k_norm.weight"])
            q = self.rope(q, pos)
            k = self.rope(k, pos)
            if self.k_cache[row][li] is not None:
                k = torch.cat([self.k_cache[row][li], k], dim=1)
                v = torch.cat([self.v_cache[row][li], v], dim=1)
            self.k_cache[row][li], self.v_cache[row][li] = k, v
            S = k.shape[1]
            g = H // Hkv
            kk = k.repeat_interleave(g, dim=0)                                      # [H, S, D]
            vv = v.repeat_interleave(g, dim=0)
            scores = (q * scale) @ kk.transpose(1, 2)                               # [H, L, S] fp32
            if L > 1:                                                               # causal, lower-right aligned
                qi = torch.arange(off, off + L)[:, None]
                kj = torch.arange(S)[None, :]
                scores = scores.masked_fill(kj > qi, float("-inf"))
            pattn = torch.softmax(scores, dim=-1)
            o = self.r(pattn @ vv)                                                  # [H, L, D]
            o = o.transpose(0, 1).reshape(L, H * D)
            h = self.r(h + self.linear(o, self.w[p + ".self_attn.o_proj.weight"]))
            x = self.rmsnorm(h, self.w[p + ".post_attention_layernorm.weight"])
            gate = self.linear(x, self.w[p + ".mlp.gate_proj.weight"])
            up = self.linear(x, self.w[p + ".mlp.up_proj.weight"])
            act = self.r(self.r(gate * self.r(torch.sigmoid(gate))) * up)
            h = self.r(h + self.linear(act, self.w[p + ".mlp.down_proj.weight"]))
        self.offset[row] = off + L
        h = self.rmsnorm(h, self.w["model.norm.weight"])
        self.last_hidden = h                          # model.norm(h): what Soprano's decoder consumes (Soprano.swift:264)
        if logit_positions is not None:
            h = h[torch.as_tensor(list(logit_positions), dtype=torch.long)]
        if head is None:
            head = self.w.get("lm_head.weight") if not cfg.tie_word_embeddings else None
            if head is None:
                head = self.w["model.embed_tokens.weight"]
        return self.linear(h, head)                                                 # [L, V]

    def forward(self, ids_per_row, logit_positions=None):
        """ids_per_row: list (len = batch) of 1-D int sequences (may differ in length).
        Returns list of logits [L_r, V] float32 (bf16-rounded values when round='bf16'); with logit_positions (one
        index list per row) only those positions' logits."""
        out = []
        with torch.no_grad():
            for r, ids in enumerate(ids_per_row):
                out.append(self._forward_row(r, torch.as_tensor(np.asarray(ids, dtype=np.int64)),
                                             None if logit_positions is None else logit_positions[r]))
        return out


def make_synthetic_weights(cfg: LlamaConfig, seed: int = 4321, dtype=torch.bfloat16) -> dict:
    """Seeded synthetic weights (mis-synth-v1, oracle/synth.py): matrices U(+-sqrt(3)*std) with
    std chosen so activations stay O(1) (1/sqrt(fan_in)); norm weights 1 + U(+-0.1); embeddings
    U(+-sqrt(3))*0.5.  Keys = HF/MLX Llama layout.  The tensor key numbering (one 64-bit key per
    tensor) is shared with the device-side generator used by bench.py (csrc/lm_engine.hip)."""
    from . import synth
    W = {}
    d, ff = cfg.hidden_size, cfg.intermediate_size
    H, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.resolved_head_dim

    def mat(key, shape, amp):
        return torch.from_numpy(synth.synth_tensor(seed * 100000 + key, shape, amp)).to(dtype)

    W["model.embed_tokens.weight"] = mat(1, (cfg.vocab_size, d), 0.5 * math.sqrt(3.0))
    W["model.norm.weight"] = (1.0 + mat(2, (d,), 0.1).float()).to(dtype)
    if not cfg.tie_word_embeddings:
        W["lm_head.weight"] = mat(3, (cfg.vocab_size, d), math.sqrt(3.0 / d) * 2.0)
    for li in range(cfg.num_hidden_layers):
        p = f"model.layers.{li}"
        k = 100 + li * 16
        W[p + ".input_layernorm.weight"] = (1.0 + mat(k + 0, (d,), 0.1).float()).to(dtype)
        W[p + ".post_attention_layernorm.weight"] = (1.0 + mat(k + 1, (d,), 0.1).float()).to(dtype)
        W[p + ".self_attn.q_proj.weight"] = mat(k + 2, (H * D, d), math.sqrt(3.0 / d) * 1.5)
        W[p + ".self_attn.k_proj.weight"] = mat(k + 3, (Hkv * D, d), math.sqrt(3.0 / d) * 1.5)
        W[p + ".self_attn.v_proj.weight"] = mat(k + 4, (Hkv * D, d), math.sqrt(3.0 / d))
        W[p + ".self_attn.o_proj.weight"] = mat(k + 5, (d, H * D), math.sqrt(3.0 / (H * D)) * 0.5)
        W[p + ".mlp.gate_proj.weight"] = mat(k + 6, (ff, d), math.sqrt(3.0 / d))
        W[p + ".mlp.up_proj.weight"] = mat(k + 7, (ff, d), math.sqrt(3.0 / d))
        W[p + ".mlp.down_proj.weight"] = mat(k + 8, (d, ff), math.sqrt(3.0 / ff) * 0.5)
        if cfg.qk_norm:
            W[p + ".self_attn.q_norm.weight"] = (1.0 + mat(k + 9, (D,), 0.1).float()).to(dtype)
            W[p + ".self_attn.k_norm.weight"] = (1.0 + mat(k + 10, (D,), 0.1).float()).to(dtype)
    return W
