"""Whisper encoder / decoder: CPU restatement of the reference.  Test infrastructure only.

Follows (Sources/MLXAudioSTT/Models/Whisper/):
  * WhisperAttention            WhisperLayers.swift:11-73    (k_proj has NO bias :29; q,v,out do)
  * WhisperEncoderLayer         WhisperLayers.swift:77-108   pre-LN block, exact-erf GELU
  * WhisperEncoder              WhisperLayers.swift:110-156  gelu(conv1 k3 p1) -> gelu(conv2 k3 s2 p1) -> +pos -> layers -> LN
  * WhisperDecoderLayer         WhisperLayers.swift:169-261  self-attn (KV cache) -> cross-attn (cached once) -> MLP
  * WhisperDecoder              WhisperLayers.swift:263-328  tok+pos embed, causal mask for prefill, tied vocab projection
  * transcribeChunk loop        WhisperModel.swift:186-282   suppress masks :293-309, greedy/temperature sample :284-291
  * HF weight layout            WhisperModel.swift:337-363   (conv weights [out, in, k] -> MLX [out, k, in])
Third-party semantics restated [3P mlx-swift]: Linear = T(x W^T + b); LayerNorm (eps 1e-5) in f32 -> T;
Conv1d NLC cross-correlation; gelu = exact erf form; SDPA f32 softmax; T = model dtype.
`round` = "bf16" rounds at every primitive boundary (the engine computes in bf16; fp16 checkpoints are
converted at load); `round` = None is plain float32 and is what the HF `transformers` WhisperModel
cross-check compares against (tests/test_oracle_whisper.py)."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as Fn


@dataclass
class WhisperConfig:           # WhisperConfig.swift:3-90 (defaults = whisper-tiny)
    vocab_size: int = 51865
    num_mel_bins: int = 80
    d_model: int = 384
    encoder_layers: int = 4
    encoder_attention_heads: int = 6
    encoder_ffn_dim: int = 1536
    max_source_positions: int = 1500
    decoder_layers: int = 4
    decoder_attention_heads: int = 6
    decoder_ffn_dim: int = 1536
    max_target_positions: int = 448


LARGE_V3 = WhisperConfig(vocab_size=51866, num_mel_bins=128, d_model=1280, encoder_layers=32,
                         encoder_attention_heads=20, encoder_ffn_dim=5120, decoder_layers=32,
                         decoder_attention_heads=20, decoder_ffn_dim=5120)
TINY = WhisperConfig(vocab_size=600, num_mel_bins=80, d_model=128, encoder_layers=2, encoder_attention_heads=2,
                     encoder_ffn_dim=256, max_source_positions=1500, decoder_layers=2, decoder_attention_heads=2,
                     decoder_ffn_dim=256, max_target_positions=448)


def _rounder(mode):
    if mode is None:
        return lambda t: t
    return lambda t: t.to(torch.bfloat16).to(torch.float32)


class WhisperOracle:
    """weights: dict in the HF layout (model.encoder.*, model.decoder.*), conv weights [out, in, k]."""

    def __init__(self, cfg: WhisperConfig, weights: dict, round: str | None = "bf16"):
        self.cfg = cfg
        self.r = _rounder(round)
        self.w = {k: torch.as_tensor(v).to(torch.float32) for k, v in weights.items()}
        self.reset(0)

    def reset(self, batch):
        L = self.cfg.decoder_layers
        self.self_k = [[None] * L for _ in range(batch)]
        self.self_v = [[None] * L for _ in range(batch)]
        self.cross_k = [[None] * L for _ in range(batch)]
        self.cross_v = [[None] * L for _ in range(batch)]
        self.pos = [0] * batch
        self.enc = [None] * batch

    # -- primitives
    def linear(self, x, p, bias=True):
        y = x @ self.w[p + ".weight"].t()
        if bias and (p + ".bias") in self.w:
            y = y + self.w[p + ".bias"]
        return self.r(y)

    def ln(self, x, p):
        return self.r(Fn.layer_norm(x, (x.shape[-1],), self.w[p + ".weight"], self.w[p + ".bias"], 1e-5))

    def gelu(self, x):
        return self.r(0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0))))

    def attention(self, q, k, v, H, mask=None):
        """q [Tq, d], k/v [Tk, d] -> [Tq, d]; f32 softmax, one rounding at the output."""
        Tq, d = q.shape
        D = d // H
        qh = q.view(Tq, H, D).transpose(0, 1)
        kh = k.view(-1, H, D).transpose(0, 1)
        vh = v.view(-1, H, D).transpose(0, 1)
        s = (qh * (D ** -0.5)) @ kh.transpose(1, 2)
        if mask is not None:
            s = s + mask
        o = self.r(torch.softmax(s, -1) @ vh)
        return o.transpose(0, 1).reshape(Tq, d)

    # -- encoder (one utterance): features [3000, n_mels] -> [1500, d]
    def encode_row(self, feats: torch.Tensor) -> torch.Tensor:
        cfg, p = self.cfg, "model.encoder"
        x = self.r(feats.to(torch.float32)).t()[None]                                      # [1, mels, T]
        h = self.r(Fn.conv1d(x, self.w[p + ".conv1.weight"], self.w[p + ".conv1.bias"], padding=1))
        h = self.gelu(h)
        h = self.r(Fn.conv1d(h, self.w[p + ".conv2.weight"], self.w[p + ".conv2.bias"], stride=2, padding=1))
        h = self.gelu(h)[0].t()                                                            # [T/2, d]
        h = self.r(h + self.w[p + ".embed_positions.weight"][: h.shape[0]])
        H = cfg.encoder_attention_heads
        for li in range(cfg.encoder_layers):
            q = f"{p}.layers.{li}"
            x = self.ln(h, q + ".self_attn_layer_norm")
            a = self.attention(self.linear(x, q + ".self_attn.q_proj"), self.linear(x, q + ".self_attn.k_proj", bias=False),
                               self.linear(x, q + ".self_attn.v_proj"), H)
            h = self.r(h + self.linear(a, q + ".self_attn.out_proj"))
            x = self.ln(h, q + ".final_layer_norm")
            x = self.gelu(self.linear(x, q + ".fc1"))
            h = self.r(h + self.linear(x, q + ".fc2"))
        return self.ln(h, p + ".layer_norm")

    def encode(self, feats_batch):
        with torch.no_grad():
            out = []
            for b, f in enumerate(feats_batch):
                e = self.encode_row(torch.as_tensor(f))
                self.enc[b] = e
                out.append(e)
            return out

    # -- decoder: Tnew tokens of row b at positions pos[b].. -> logits [Tnew, V]
    def decode_row(self, b: int, tokens) -> torch.Tensor:
        cfg, p = self.cfg, "model.decoder"
        tok = torch.as_tensor(np.asarray(tokens, np.int64))
        T, start = tok.shape[0], self.pos[b]
        h = self.r(self.w[p + ".embed_tokens.weight"][tok] + self.w[p + ".embed_positions.weight"][start:start + T])
        H = cfg.decoder_attention_heads
        mask = None
        if T > 1:                                         # causalMask, WhisperLayers.swift:311-319
            rows = torch.arange(start, start + T)[:, None]
            cols = torch.arange(start + T)[None, :]
            mask = torch.where(cols <= rows, 0.0, -1e9)
        enc = self.enc[b]
        for li in range(cfg.decoder_layers):
            q = f"{p}.layers.{li}"
            x = self.ln(h, q + ".self_attn_layer_norm")
            k = self.linear(x, q + ".self_attn.k_proj", bias=False)
            v = self.linear(x, q + ".self_attn.v_proj")
            if self.self_k[b][li] is not None:
                k = torch.cat([self.self_k[b][li], k], 0)
                v = torch.cat([self.self_v[b][li], v], 0)
            self.self_k[b][li], self.self_v[b][li] = k, v
            a = self.attention(self.linear(x, q + ".self_attn.q_proj"), k, v, H, mask)
            h = self.r(h + self.linear(a, q + ".self_attn.out_proj"))
            x = self.ln(h, q + ".encoder_attn_layer_norm")
            if self.cross_k[b][li] is None:
                self.cross_k[b][li] = self.linear(enc, q + ".encoder_attn.k_proj", bias=False)
                self.cross_v[b][li] = self.linear(enc, q + ".encoder_attn.v_proj")
            a = self.attention(self.linear(x, q + ".encoder_attn.q_proj"), self.cross_k[b][li], self.cross_v[b][li], H)
            h = self.r(h + self.linear(a, q + ".encoder_attn.out_proj"))
            x = self.ln(h, q + ".final_layer_norm")
            x = self.gelu(self.linear(x, q + ".fc1"))
            h = self.r(h + self.linear(x, q + ".fc2"))
        self.pos[b] = start + T
        h = self.ln(h, p + ".layer_norm")
        return self.r(h @ self.w[p + ".embed_tokens.weight"].t())        # projectToVocab (tied), :325-327

    def decode(self, tokens_per_row):
        with torch.no_grad():
            return [self.decode_row(b, t) for b, t in enumerate(tokens_per_row)]


def apply_suppress(logits: np.ndarray, step: int, begin_suppress, suppress, timestamp_begin: int) -> np.ndarray:
    """WhisperModel.swift:228-236,293-309: additive -1e9 masks (begin_suppress at step 0 only)."""
    out = np.array(logits, np.float32, copy=True)
    if step == 0:
        for i in begin_suppress:
            if 0 <= i < out.shape[0]:
                out[i] += np.float32(-1e9)
    for i in suppress:
        if 0 <= i < out.shape[0]:
            out[i] += np.float32(-1e9)
    if timestamp_begin < out.shape[0]:
        out[timestamp_begin:] += np.float32(-1e9)
    return out


def make_synthetic_weights(cfg: WhisperConfig, seed: int = 777, dtype=torch.bfloat16) -> dict:
    """Seeded synthetic weights (mis-synth-v1) in the HF key layout; matrices U(+-g*sqrt(3/fan_in)), LN weights
    1 + U(+-0.1), biases U(+-0.05), embeddings U(+-0.5)."""
    from . import synth
    W, key = {}, [seed * 100000]

    def t(shape, amp):
        key[0] += 1
        return torch.from_numpy(synth.synth_tensor(key[0], shape, amp))

    def lin(p, out_f, in_f, bias=True, gain=1.0):
        W[p + ".weight"] = t((out_f, in_f), gain * math.sqrt(3.0 / in_f)).to(dtype)
        if bias:
            W[p + ".bias"] = t((out_f,), 0.05).to(dtype)

    def lnp(p, d):
        W[p + ".weight"] = (1.0 + t((d,), 0.1)).to(dtype)
        W[p + ".bias"] = t((d,), 0.05).to(dtype)

    def attn(p, d):
        lin(p + ".q_proj", d, d); lin(p + ".k_proj", d, d, bias=False); lin(p + ".v_proj", d, d)
        lin(p + ".out_proj", d, d, gain=0.5)

    d = cfg.d_model
    e = "model.encoder"
    W[e + ".conv1.weight"] = t((d, cfg.num_mel_bins, 3), math.sqrt(3.0 / (3 * cfg.num_mel_bins))).to(dtype)
    W[e + ".conv1.bias"] = t((d,), 0.05).to(dtype)
    W[e + ".conv2.weight"] = t((d, d, 3), math.sqrt(3.0 / (3 * d))).to(dtype)
    W[e + ".conv2.bias"] = t((d,), 0.05).to(dtype)
    W[e + ".embed_positions.weight"] = t((cfg.max_source_positions, d), 0.3).to(dtype)
    for li in range(cfg.encoder_layers):
        q = f"{e}.layers.{li}"
        attn(q + ".self_attn", d); lnp(q + ".self_attn_layer_norm", d)
        lin(q + ".fc1", cfg.encoder_ffn_dim, d); lin(q + ".fc2", d, cfg.encoder_ffn_dim, gain=0.5)
        lnp(q + ".final_layer_norm", d)
    lnp(e + ".layer_norm", d)
    dd = "model.decoder"
    W[dd + ".embed_tokens.weight"] = t((cfg.vocab_size, d), 0.5).to(dtype)
    W[dd + ".embed_positions.weight"] = t((cfg.max_target_positions, d), 0.3).to(dtype)
    for li in range(cfg.decoder_layers):
        q = f"{dd}.layers.{li}"
        attn(q + ".self_attn", d); lnp(q + ".self_attn_layer_norm", d)
        attn(q + ".encoder_attn", d); lnp(q + ".encoder_attn_layer_norm", d)
        lin(q + ".fc1", cfg.decoder_ffn_dim, d); lin(q + ".fc2", d, cfg.decoder_ffn_dim, gain=0.5)
        lnp(q + ".final_layer_norm", d)
    lnp(dd + ".layer_norm", d)
    return W
