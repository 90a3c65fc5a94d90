"""Qwen3-TTS (talker + code predictor + speech-tokenizer decoder): CPU restatement of the reference.
Test infrastructure only.

Follows (Sources/MLXAudioTTS/Models/Qwen3TTS/):
  * generateVoiceDesign frame loop          Qwen3TTS.swift:306-569   (talker step -> code 0 -> 15 code-predictor steps
                                            -> next input = text embed + sum of 16 code embeddings; EOS on code 0)
  * prepareGenerationInputs                 Qwen3TTS.swift:883-1000  (every prefill position is text_proj(text_emb[t]) and/or
                                            codec_emb[c]: expressed here as (text_id, codec_id) pairs, -1 = absent)
  * sampleToken                             Qwen3TTS.swift:1003-1118
  * Talker / TalkerAttention / MRoPE        Qwen3TTSTalker.swift:41-305 (all three position streams are equal for TTS, so the
                                            interleaved MRoPE reduces to plain RoPE(theta); rotation written as array ops in T)
  * Qwen3TTSCodePredictor                   Qwen3TTSCodePredictor.swift:12-243
  * Qwen3TTSSpeechTokenizerDecoder          Qwen3TTSSpeechTokenizer.swift:9-790,888-946 (+ EuclideanCodebook
                                            Sources/MLXAudioCodecs/Mimi/Quantization.swift:7-60)
The decoder is causal end to end, so `streamingStep` over chunks (what decodeChunk uses, Qwen3TTS.swift:214-231) equals
one full-sequence call; this file implements the full-sequence form (float32).
Third-party semantics restated [3P]: MLXNN.Conv1d NLC weight [out, k, in]; MLXNN.ConvTransposed1d weight [out, k, in],
y[n*s + j] += x[n, c] * w[o, j, c]; Linear with bias is one fused op (single rounding); Embedding gather.
Sampler: deterministic realisation ("mis-sampler-v1" machinery, oracle/sampler.py) of sampleToken's set semantics:
suppress -> repetition penalty over the UNIQUE generated ids (bf16 arithmetic) -> top-k (ties at the k-th value kept whole)
-> top-p on softmax of the filtered logits at temperature 1 (exact integer masses) -> min-p -> EOS logit restored ->
categorical(T(l / T(temp))) by inverse CDF."""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as TF

from . import sampler as osamp
from . import synth
from .llama import LlamaConfig, LlamaOracle
from .llama import make_synthetic_weights as llama_synth

F = np.float32


@dataclass
class DecoderConfig:                         # Qwen3TTSTokenizerDecoderConfig, Qwen3TTSConfig.swift:307-385
    latent_dim: int = 1024
    codebook_dim: int = 512
    codebook_size: int = 2048
    decoder_dim: int = 1536
    hidden_size: int = 512
    intermediate_size: int = 1024
    head_dim: int = 64
    num_attention_heads: int = 16
    num_hidden_layers: int = 8
    num_key_value_heads: int = 16
    num_quantizers: int = 16
    num_semantic_quantizers: int = 1
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    upsample_rates: tuple = (8, 5, 4, 3)
    upsampling_ratios: tuple = (2, 2)

    @property
    def total_upsample(self) -> int:
        return int(np.prod(self.upsample_rates) * np.prod(self.upsampling_ratios))


@dataclass
class Qwen3TTSConfig:
    talker: LlamaConfig = field(default_factory=lambda: LlamaConfig(
        hidden_size=1024, num_hidden_layers=28, intermediate_size=3072, num_attention_heads=16, num_key_value_heads=8,
        head_dim=128, rms_norm_eps=1e-6, vocab_size=3072, rope_theta=1e6, rope_scaling=None, tie_word_embeddings=False,
        qk_norm=True, rope_plain=True, rope_ops_in_dtype=True))
    predictor: LlamaConfig = field(default_factory=lambda: LlamaConfig(
        hidden_size=1024, num_hidden_layers=5, intermediate_size=3072, num_attention_heads=16, num_key_value_heads=8,
        head_dim=128, rms_norm_eps=1e-6, vocab_size=2048, rope_theta=1e6, rope_scaling=None, tie_word_embeddings=False,
        qk_norm=True, rope_plain=True, rope_ops_in_dtype=True))
    num_code_groups: int = 16
    text_hidden_size: int = 2048
    text_vocab_size: int = 151936
    codec_eos_token_id: int = 2150
    codec_think_id: int = 2154
    codec_nothink_id: int = 2155
    codec_think_bos_id: int = 2156
    codec_think_eos_id: int = 2157
    codec_pad_id: int = 2148
    codec_bos_id: int = 2149
    tts_pad_token_id: int = 151671
    tts_bos_token_id: int = 151672
    tts_eos_token_id: int = 151673
    decoder: DecoderConfig = field(default_factory=DecoderConfig)


def _tiny_lm(layers, vocab, hidden=256, ff=384):
    return LlamaConfig(hidden_size=hidden, num_hidden_layers=layers, intermediate_size=ff, num_attention_heads=4,
                       num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-6, vocab_size=vocab, rope_theta=1e6,
                       rope_scaling=None, tie_word_embeddings=False, qk_norm=True, rope_plain=True, rope_ops_in_dtype=True)


TINY = Qwen3TTSConfig(
    # talker vocabulary = 96 audio codes (== decoder codebook_size) + 1024 suppressed special ids
    talker=_tiny_lm(2, 1120), predictor=_tiny_lm(2, 96), num_code_groups=4, text_hidden_size=128, text_vocab_size=500,
    codec_eos_token_id=1030, codec_think_id=1034, codec_nothink_id=1035, codec_think_bos_id=1036, codec_think_eos_id=1037,
    codec_pad_id=1028, codec_bos_id=1029, tts_pad_token_id=491, tts_bos_token_id=492, tts_eos_token_id=493,
    decoder=DecoderConfig(latent_dim=64, codebook_dim=32, codebook_size=96, decoder_dim=96, hidden_size=64,
                          intermediate_size=96, head_dim=16, num_attention_heads=4, num_hidden_layers=2,
                          num_key_value_heads=4, num_quantizers=4, num_semantic_quantizers=1, upsample_rates=(3, 2),
                          upsampling_ratios=(2,)))
# talker hidden != predictor hidden exercises small_to_mtp_projection (Qwen3TTSCodePredictor.swift:207-211)
TINY_PROJ = Qwen3TTSConfig(**{**TINY.__dict__, "predictor": _tiny_lm(2, 96, hidden=128, ff=256)})


# ------------------------------------------------------------------------------------------------ sampler
def sample_token(logits, temperature, top_p, top_k, penalty, generated, suppress, eos_id, min_p, seed, row, step):
    """logits [V] float32 holding bf16 values.  suppress: iterable of ids or None.  Returns the token id."""
    bf = synth.bf16_round
    l = np.array(logits, F, copy=True)
    V = l.shape[0]
    if suppress is not None:
        l[np.asarray(list(suppress), np.int64)] = -np.inf
    if generated is not None and len(generated) and penalty != 1.0:
        pen = bf(np.asarray([penalty], F))[0]
        for t in sorted(set(int(t) for t in generated)):
            if t < V:
                l[t] = bf(np.asarray([l[t] * pen if l[t] < 0 else l[t] / pen], F))[0]
    if temperature <= 0:
        return int(np.argmax(l))
    eos_logit = l[eos_id] if eos_id is not None and 0 <= eos_id < V else None
    f = l.copy()
    if 0 < top_k < V:
        kth = np.sort(f)[::-1][top_k - 1]
        f[f < kth] = -np.inf                          # tie group at the k-th value kept whole
    if 0.0 < top_p < 1.0:                             # nucleus on softmax(f) at temperature 1
        m = f.max()
        e = osamp.det_exp(np.clip((f - m).astype(F), F(-100.0), F(0.0)))
        e = np.where(np.isfinite(f), e, F(0.0)).astype(F)
        E = (e.astype(np.float64) * float(2 ** 40)).astype(np.uint64)
        Z = int(E.sum(dtype=np.uint64))
        thr = int(np.uint64(np.float64(F(1.0) - F(top_p)) * np.float64(Z)))
        order = np.argsort(f, kind="stable")
        vals = f[order]
        cum = np.cumsum(E[order], dtype=np.uint64)
        # keep value groups whose cumulative mass (ascending, whole tie group included) exceeds the threshold
        keep_sorted = np.zeros(V, bool)
        i = 0
        while i < V:
            j = i
            while j + 1 < V and vals[j + 1] == vals[i]:
                j += 1
            if int(cum[j]) > thr:
                keep_sorted[i:j + 1] = True
            i = j + 1
        keep = np.zeros(V, bool)
        keep[order] = keep_sorted
        f = np.where(keep, f, F(-np.inf)).astype(F)
    if min_p > 0.0:
        cut = bf(np.asarray([F(math.log(min_p))], F))[0]
        lim = bf(np.asarray([f.max() + cut], F))[0]
        f = np.where(f < lim, F(-np.inf), f).astype(F)
    if eos_logit is not None:
        f[eos_id] = eos_logit
    tb = bf(np.asarray([temperature], F))[0]
    x = bf((f / tb).astype(F))                        # categorical(filteredLogits / temperature), array op in bf16
    m = x.max()
    e = osamp.det_exp(np.clip((x - m).astype(F), F(-100.0), F(0.0)))
    e = np.where(np.isfinite(x), e, F(0.0)).astype(F)
    E = (e.astype(np.float64) * float(2 ** 40)).astype(np.uint64)
    Z = int(E.sum(dtype=np.uint64))
    r = (osamp.rand64(seed, row, step) * Z) >> 64
    return int(np.searchsorted(np.cumsum(E, dtype=np.uint64), np.uint64(r), side="right"))


# ------------------------------------------------------------------------------------------------ LM side
class Qwen3TTSOracle:
    """weights: keys after Qwen3TTSTalkerForConditionalGeneration.sanitize (no "talker." prefix)."""

    def __init__(self, cfg: Qwen3TTSConfig, weights: dict):
        self.cfg = cfg
        self.w = {k: torch.as_tensor(v).to(torch.float32) for k, v in weights.items()}
        tw = {k: v for k, v in self.w.items() if k.startswith("model.")}
        tw["model.embed_tokens.weight"] = self.w["model.codec_embedding.weight"]
        tw["lm_head.weight"] = self.w["codec_head.weight"]
        self.talker = LlamaOracle(cfg.talker, tw, round="bf16")
        pw = {k[len("code_predictor."):]: v for k, v in self.w.items() if k.startswith("code_predictor.model.")}
        pw["model.embed_tokens.weight"] = torch.zeros(1, cfg.predictor.hidden_size)
        pw["lm_head.weight"] = self.w["code_predictor.lm_head.0.weight"]
        self.pred = LlamaOracle(cfg.predictor, pw, round="bf16")
        self.r = self.talker.r

    def text_embed(self, ids):
        """textProjection(textEmbedding(ids)): ResizeMLP fc2(silu(fc1(x))) with biases (Qwen3TTSTalker.swift:212-225)."""
        w, r = self.w, self.r
        x = w["model.text_embedding.weight"][torch.as_tensor(np.asarray(ids, np.int64))]
        h = r(x @ w["text_projection.linear_fc1.weight"].t() + w["text_projection.linear_fc1.bias"])
        h = r(h * r(torch.sigmoid(h)))
        return r(h @ w["text_projection.linear_fc2.weight"].t() + w["text_projection.linear_fc2.bias"])

    def codec_embed(self, ids):
        return self.w["model.codec_embedding.weight"][torch.as_tensor(np.asarray(ids, np.int64))]

    def codec_embed_icl(self, ref_codes):
        """codecEmbedIcl (Qwen3TTS.swift:249-262) without its codec_bos row: reference codes [n_q, T] -> [T, d], the talker's codec
        embedding of code 0 plus the code predictor's embedding i of code i + 1, one model-dtype rounding per add."""
        rc = np.asarray(ref_codes, np.int64)
        e = self.codec_embed(rc[0])
        for i in range(min(self.cfg.num_code_groups - 1, rc.shape[0] - 1)):
            e = self.r(e + self.w[f"code_predictor.model.codec_embedding.{i}.weight"][torch.as_tensor(rc[i + 1])])
        return e

    def position_embeds(self, text_ids, codec_ids, extra_rows=None):
        """Prefill positions as (text id | -1, codec id | -1) pairs -> [P, d] bf16-valued.  Codec ids past the vocabulary index
        `extra_rows` ([n, d]: speaker vector / codecEmbedIcl rows of the in-context prompt, prepareICLGenerationInputs :753-837)."""
        out = []
        V = self.cfg.talker.vocab_size
        for t, c in zip(text_ids, codec_ids):
            e = None
            if t >= 0:
                e = self.text_embed([t])[0]
            if c >= 0:
                ce = self.codec_embed([c])[0] if c < V else self.r(torch.as_tensor(extra_rows[c - V], dtype=torch.float32))
                e = ce if e is None else self.r(e + ce)
            out.append(e)
        return torch.stack(out)

    def predictor_codes(self, hidden, code0, params, row, frame, forced=None, want_logits=False):
        """15 (num_code_groups - 1) sequential code-predictor steps for one frame (Qwen3TTS.swift:431-461)."""
        cfg, w = self.cfg, self.w
        self.pred.reset(1)
        proj = "code_predictor.small_to_mtp_projection.weight" in w

        def pin(e):
            if proj:
                return self.r(e @ w["code_predictor.small_to_mtp_projection.weight"].t() + w["code_predictor.small_to_mtp_projection.bias"])
            return e
        codes, all_logits = [int(code0)], []
        for i in range(cfg.num_code_groups - 1):
            if i == 0:
                x = torch.stack([hidden, self.codec_embed([code0])[0]])
            else:
                x = w[f"code_predictor.model.codec_embedding.{i - 1}.weight"][codes[-1]][None]
            lg = self.pred.forward_embeds(0, pin(x), head=w[f"code_predictor.lm_head.{i}.weight"])[-1].numpy()
            all_logits.append(lg)
            if forced is not None:
                tok = int(forced[i + 1])
            else:
                tok = sample_token(lg, params["temperature"], params["top_p"], params["top_k"], 1.0, None, None, None,
                                   params["min_p"], params["seed"], row, frame * cfg.num_code_groups + i + 1)
            codes.append(tok)
        return (codes, all_logits) if want_logits else codes

    def next_input(self, text_embed_row, codes):
        """text + codec_emb(code0) + sum_i predictor.codec_embedding[i](code_{i+1}), sequential bf16 adds (:475-480)."""
        e = self.codec_embed([codes[0]])[0]
        for i, c in enumerate(codes[1:]):
            e = self.r(e + self.w[f"code_predictor.model.codec_embedding.{i}.weight"][c])
        return self.r(text_embed_row + e)

    def generate_row(self, text_ids, codec_ids, trailing_text_ids, params, row=0, max_frames=16, forced_codes=None, extra_rows=None):
        """One utterance.  Returns (codes [n_frames, G] int, per-frame talker logits).  `forced_codes` [n, G] teacher-forces
        the sampled values (the logits are still the oracle's)."""
        cfg = self.cfg
        self.talker.reset(1)
        suppress = [t for t in range(cfg.talker.vocab_size - 1024, cfg.talker.vocab_size) if t != cfg.codec_eos_token_id]
        x = self.position_embeds(text_ids, codec_ids, extra_rows)
        trailing = self.text_embed(trailing_text_ids) if len(trailing_text_ids) else torch.zeros(0, cfg.talker.hidden_size)
        pad = self.text_embed([cfg.tts_pad_token_id])[0]
        frames, tlogits, gen0 = [], [], []
        for step in range(max_frames):
            lg = self.talker.forward_embeds(0, x)[-1].numpy()
            hidden = self.talker.last_hidden[-1]
            tlogits.append(lg)
            if forced_codes is not None:
                if step >= len(forced_codes):
                    break
                code0 = int(forced_codes[step][0])
            else:
                code0 = sample_token(lg, params["temperature"], params["top_p"], params["top_k"], params["repetition_penalty"],
                                     gen0, suppress, cfg.codec_eos_token_id, params["min_p"], params["seed"], row,
                                     step * cfg.num_code_groups)
            if code0 == cfg.codec_eos_token_id:
                break
            codes = self.predictor_codes(hidden, code0, params, row, step,
                                         forced=None if forced_codes is None else forced_codes[step])
            te = trailing[step] if step < trailing.shape[0] else pad
            x = self.next_input(te, codes)[None]
            gen0.append(code0)
            frames.append(codes)
        return np.asarray(frames, np.int32).reshape(-1, cfg.num_code_groups), tlogits


# ------------------------------------------------------------------------------------------------ speech-tokenizer decoder
def _t(a):
    return torch.as_tensor(np.asarray(a, F))


def causal_conv1d(x, w, b, dilation=1, groups=1):
    """CausalConv1d, stride 1 (Qwen3TTSSpeechTokenizer.swift:132-196): left pad (k-1)*dilation.  x [B,C,T]; w [Co,k,Ci/g]."""
    k = w.shape[1]
    xp = TF.pad(x, ((k - 1) * dilation, 0))
    return TF.conv1d(xp, w.permute(0, 2, 1).contiguous(), b, dilation=dilation, groups=groups)


def causal_conv_transpose1d(x, w, b, stride):
    """ConvTransposed1d(padding 0) then drop the last k - stride samples (:533-551, :732-749).  w [Co,k,Ci]."""
    k = w.shape[1]
    y = TF.conv_transpose1d(x, w.permute(2, 0, 1).contiguous(), b, stride=stride)
    return y[..., : y.shape[-1] - (k - stride)] if k > stride else y


def snake_beta(x, alpha, beta):
    a, b = torch.exp(alpha)[None, :, None], torch.exp(beta)[None, :, None]
    s = torch.sin(x * a)
    return x + (1.0 / (b + 1e-9)) * s * s


class SpeechDecoderOracle:
    def __init__(self, cfg: DecoderConfig, weights: dict):
        self.cfg = cfg
        self.w = {k: _t(v) for k, v in weights.items()}

    def quantizer_decode(self, codes):
        """SplitResidualVectorQuantizer.decode (:91-118): codes [B, nq, T] -> [B, codebook_dim, T]."""
        cfg, w = self.cfg, self.w
        codes = torch.as_tensor(np.asarray(codes, np.int64))
        out = 0
        for name, lo, hi in (("rvq_first", 0, cfg.num_semantic_quantizers), ("rvq_rest", cfg.num_semantic_quantizers, cfg.num_quantizers)):
            q = 0
            for i in range(lo, hi):
                p = f"decoder.quantizer.{name}.vq.layers.{i - lo}.codebook"
                emb = w[p + ".embedding_sum"] / torch.clamp(w[p + ".cluster_usage"], min=1e-5)[:, None]
                q = q + emb[codes[:, i]].transpose(1, 2)                       # [B, dim, T]
            pw = w[f"decoder.quantizer.{name}.output_proj.weight"][:, 0, :]       # Conv1d k=1, no bias: [out, in]
            out = out + torch.einsum("oi,bit->bot", pw, q)
        return out

    def rmsnorm(self, x, wt):
        return wt * (x * torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + self.cfg.rms_norm_eps))

    def transformer(self, x):
        """DecoderTransformer (:449-503), x [B, T, latent] -> [B, T, latent]; causal, full context."""
        cfg, w, P = self.cfg, self.w, "decoder.pre_transformer"
        B, T, _ = x.shape
        H, D = cfg.num_attention_heads, cfg.head_dim
        x = x @ w[P + ".input_proj.weight"].t() + w[P + ".input_proj.bias"]
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
        ang = torch.arange(T, dtype=torch.float32)[:, None] * inv[None]
        cos, sin = torch.cat([torch.cos(ang)] * 2, -1), torch.cat([torch.sin(ang)] * 2, -1)

        def rot(t):
            return torch.cat([-t[..., D // 2:], t[..., : D // 2]], -1)
        mask = torch.full((T, T), float("-inf")).triu(1)
        for i in range(cfg.num_hidden_layers):
            p = f"{P}.layers.{i}"
            h = self.rmsnorm(x, w[p + ".input_layernorm.weight"])
            q = (h @ w[p + ".self_attn.q_proj.weight"].t()).view(B, T, H, D).transpose(1, 2)
            k = (h @ w[p + ".self_attn.k_proj.weight"].t()).view(B, T, cfg.num_key_value_heads, D).transpose(1, 2)
            v = (h @ w[p + ".self_attn.v_proj.weight"].t()).view(B, T, cfg.num_key_value_heads, D).transpose(1, 2)
            q, k = q * cos + rot(q) * sin, k * cos + rot(k) * sin
            g = H // cfg.num_key_value_heads
            k, v = k.repeat_interleave(g, 1), v.repeat_interleave(g, 1)
            a = torch.softmax((q * D ** -0.5) @ k.transpose(-1, -2) + mask, -1) @ v
            a = a.transpose(1, 2).reshape(B, T, H * D) @ w[p + ".self_attn.o_proj.weight"].t()
            x = x + w[p + ".self_attn_layer_scale.scale"] * a
            h = self.rmsnorm(x, w[p + ".post_attention_layernorm.weight"])
            m = (TF.silu(h @ w[p + ".mlp.gate_proj.weight"].t()) * (h @ w[p + ".mlp.up_proj.weight"].t())) @ w[p + ".mlp.down_proj.weight"].t()
            x = x + w[p + ".mlp_layer_scale.scale"] * m
        x = self.rmsnorm(x, w[P + ".norm.weight"])
        return x @ w[P + ".output_proj.weight"].t() + w[P + ".output_proj.bias"]

    def convnext(self, x, p):
        w = self.w
        h = causal_conv1d(x, w[p + ".dwconv.conv.weight"], w[p + ".dwconv.conv.bias"], groups=x.shape[1]).transpose(1, 2)
        h = TF.layer_norm(h, (h.shape[-1],), w[p + ".norm.weight"], w[p + ".norm.bias"], 1e-6)
        h = TF.gelu(h @ w[p + ".pwconv1.weight"].t() + w[p + ".pwconv1.bias"])
        h = w[p + ".gamma"] * (h @ w[p + ".pwconv2.weight"].t() + w[p + ".pwconv2.bias"])
        return x + h.transpose(1, 2)

    def decode(self, codes, stop_after=None):
        """Qwen3TTSSpeechTokenizerDecoder.callAsFunction (:926-946): codes [B, nq, T] int -> wav [B, T*total_upsample]."""
        cfg, w = self.cfg, self.w
        with torch.no_grad():
            h = self.quantizer_decode(codes)
            if stop_after == "quantizer":
                return h.numpy()
            h = causal_conv1d(h, w["decoder.pre_conv.conv.weight"], w["decoder.pre_conv.conv.bias"])
            h = self.transformer(h.transpose(1, 2)).transpose(1, 2)
            if stop_after == "transformer":
                return h.numpy()
            for i, f in enumerate(cfg.upsampling_ratios):
                p = f"decoder.upsample.{i}.layers"
                h = causal_conv_transpose1d(h, w[p + ".0.conv.weight"], w[p + ".0.conv.bias"], f)
                h = self.convnext(h, p + ".1")
            if stop_after == "upsample":
                return h.numpy()
            h = causal_conv1d(h, w["decoder.decoder.0.conv.weight"], w["decoder.decoder.0.conv.bias"])
            for bi, rate in enumerate(cfg.upsample_rates):
                p = f"decoder.decoder.{bi + 1}.block"
                h = snake_beta(h, w[p + ".0.alpha"], w[p + ".0.beta"])
                h = causal_conv_transpose1d(h, w[p + ".1.conv.weight"], w[p + ".1.conv.bias"], rate)
                for ri, dil in enumerate((1, 3, 9)):
                    q = f"{p}.{ri + 2}"
                    t = snake_beta(h, w[q + ".act1.alpha"], w[q + ".act1.beta"])
                    t = causal_conv1d(t, w[q + ".conv1.conv.weight"], w[q + ".conv1.conv.bias"], dilation=dil)
                    t = snake_beta(t, w[q + ".act2.alpha"], w[q + ".act2.beta"])
                    h = h + causal_conv1d(t, w[q + ".conv2.conv.weight"], w[q + ".conv2.conv.bias"])
                if stop_after == f"block{bi}":
                    return h.numpy()
            n = len(cfg.upsample_rates)
            h = snake_beta(h, w[f"decoder.decoder.{n + 1}.alpha"], w[f"decoder.decoder.{n + 1}.beta"])
            h = causal_conv1d(h, w[f"decoder.decoder.{n + 2}.conv.weight"], w[f"decoder.decoder.{n + 2}.conv.bias"])
            return torch.clamp(h, -1.0, 1.0)[:, 0].numpy()


    # -------------------------------------------------------------------------------------------- streaming (carried state)
    def reset_streaming_state(self):
        """resetStreamingState (Qwen3TTSSpeechTokenizer.swift:948-968)."""
        self._sb = {}            # CausalConv1d.streamBuffer / DecoderInitialConv / DecoderOutputConv buffers, by layer name
        self._ov = {}            # DecoderBlockUpsample.overflow, by layer name
        self._kv = None          # transformerCache: per layer (k [B,Hkv,T,D] rotated, v)

    def _conv_step(self, name, x, w, b, dilation=1, groups=1):
        """CausalConv1d.step (:199-227) and the k7 conv steps (:655-667,:710-722): prepend the carried buffer (zeros on the first
        chunk), keep the last `padding` columns, run the unpadded conv."""
        k = w.shape[1]
        pad = (k - 1) * dilation
        if pad > 0:
            buf = self._sb.get(name)
            x = torch.cat([buf, x], -1) if buf is not None else TF.pad(x, (pad, 0))
            self._sb[name] = x[..., max(0, x.shape[-1] - pad):]
        return TF.conv1d(x, w.permute(0, 2, 1).contiguous(), b, dilation=dilation, groups=groups)

    def _upsample_step(self, name, x, w, b, stride):
        """DecoderBlockUpsample.step (:553-576): full transposed conv, add the carried overflow to the head, split off the last
        k - stride outputs as the next overflow.  Both summands carry the bias - the reference counts it twice there."""
        k = w.shape[1]
        h = TF.conv_transpose1d(x, w.permute(2, 0, 1).contiguous(), b, stride=stride)
        ov = self._ov.get(name)
        if ov is not None:
            n = ov.shape[-1]
            head = h[..., :n] + ov
            h = torch.cat([head, h[..., n:]], -1) if n < h.shape[-1] else head
        trim = k - stride
        if trim > 0:
            split = max(0, h.shape[-1] - trim)
            self._ov[name] = h[..., split:]
            h = h[..., :split]
        else:
            self._ov[name] = None
        return h

    def _transformer_step(self, x):
        """DecoderTransformer with a KVCacheSimple per layer (:470-496): positions continue at the cache offset, the new queries
        see every cached key plus the causal part of the chunk."""
        cfg, w, P = self.cfg, self.w, "decoder.pre_transformer"
        B, T, _ = x.shape
        H, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        if self._kv is None:
            self._kv = [None] * cfg.num_hidden_layers
        off = 0 if self._kv[0] is None else self._kv[0][0].shape[2]
        x = x @ w[P + ".input_proj.weight"].t() + w[P + ".input_proj.bias"]
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
        ang = torch.arange(off, off + T, dtype=torch.float32)[:, None] * inv[None]
        cos, sin = torch.cat([torch.cos(ang)] * 2, -1), torch.cat([torch.sin(ang)] * 2, -1)

        def rot(t):
            return torch.cat([-t[..., D // 2:], t[..., : D // 2]], -1)
        mask = torch.full((T, off + T), float("-inf")).triu(off + 1)
        for i in range(cfg.num_hidden_layers):
            p = f"{P}.layers.{i}"
            h = self.rmsnorm(x, w[p + ".input_layernorm.weight"])
            q = (h @ w[p + ".self_attn.q_proj.weight"].t()).view(B, T, H, D).transpose(1, 2)
            k = (h @ w[p + ".self_attn.k_proj.weight"].t()).view(B, T, Hkv, D).transpose(1, 2)
            v = (h @ w[p + ".self_attn.v_proj.weight"].t()).view(B, T, Hkv, D).transpose(1, 2)
            q, k = q * cos + rot(q) * sin, k * cos + rot(k) * sin
            if self._kv[i] is not None:
                k, v = torch.cat([self._kv[i][0], k], 2), torch.cat([self._kv[i][1], v], 2)
            self._kv[i] = (k, v)
            g = H // Hkv
            kk, vv = k.repeat_interleave(g, 1), v.repeat_interleave(g, 1)
            a = torch.softmax((q * D ** -0.5) @ kk.transpose(-1, -2) + mask, -1) @ vv
            a = a.transpose(1, 2).reshape(B, T, H * D) @ w[p + ".self_attn.o_proj.weight"].t()
            x = x + w[p + ".self_attn_layer_scale.scale"] * a
            h = self.rmsnorm(x, w[p + ".post_attention_layernorm.weight"])
            m = (TF.silu(h @ w[p + ".mlp.gate_proj.weight"].t()) * (h @ w[p + ".mlp.up_proj.weight"].t())) @ w[p + ".mlp.down_proj.weight"].t()
            x = x + w[p + ".mlp_layer_scale.scale"] * m
        x = self.rmsnorm(x, w[P + ".norm.weight"])
        return x @ w[P + ".output_proj.weight"].t() + w[P + ".output_proj.bias"]

    def streaming_step(self, codes):
        """streamingStep (:971-1006): codes [B, nq, Tn] = only the NEW frames -> wav [B, Tn * total_upsample].  Call
        reset_streaming_state() first."""
        cfg, w = self.cfg, self.w
        with torch.no_grad():
            h = self.quantizer_decode(codes)
            h = self._conv_step("pre_conv", h, w["decoder.pre_conv.conv.weight"], w["decoder.pre_conv.conv.bias"])
            h = self._transformer_step(h.transpose(1, 2)).transpose(1, 2)
            for i, f in enumerate(cfg.upsampling_ratios):
                p = f"decoder.upsample.{i}.layers"
                h = causal_conv_transpose1d(h, w[p + ".0.conv.weight"], w[p + ".0.conv.bias"], f)     # k == stride: stateless (:772-779)
                q = p + ".1"
                t = self._conv_step(q, h, w[q + ".dwconv.conv.weight"], w[q + ".dwconv.conv.bias"], groups=h.shape[1]).transpose(1, 2)
                t = TF.layer_norm(t, (t.shape[-1],), w[q + ".norm.weight"], w[q + ".norm.bias"], 1e-6)
                t = TF.gelu(t @ w[q + ".pwconv1.weight"].t() + w[q + ".pwconv1.bias"])
                t = w[q + ".gamma"] * (t @ w[q + ".pwconv2.weight"].t() + w[q + ".pwconv2.bias"])
                h = h + t.transpose(1, 2)
            h = self._conv_step("dec0", h, w["decoder.decoder.0.conv.weight"], w["decoder.decoder.0.conv.bias"])
            for bi, rate in enumerate(cfg.upsample_rates):
                p = f"decoder.decoder.{bi + 1}.block"
                h = snake_beta(h, w[p + ".0.alpha"], w[p + ".0.beta"])
                h = self._upsample_step(p + ".1", h, w[p + ".1.conv.weight"], w[p + ".1.conv.bias"], rate)
                for ri, dil in enumerate((1, 3, 9)):
                    q = f"{p}.{ri + 2}"
                    t = snake_beta(h, w[q + ".act1.alpha"], w[q + ".act1.beta"])
                    t = self._conv_step(q + ".conv1", t, w[q + ".conv1.conv.weight"], w[q + ".conv1.conv.bias"], dilation=dil)
                    t = snake_beta(t, w[q + ".act2.alpha"], w[q + ".act2.beta"])
                    h = h + self._conv_step(q + ".conv2", t, w[q + ".conv2.conv.weight"], w[q + ".conv2.conv.bias"])
            n = len(cfg.upsample_rates)
            h = snake_beta(h, w[f"decoder.decoder.{n + 1}.alpha"], w[f"decoder.decoder.{n + 1}.beta"])
            h = self._conv_step("out", h, w[f"decoder.decoder.{n + 2}.conv.weight"], w[f"decoder.decoder.{n + 2}.conv.bias"])
            return torch.clamp(h, -1.0, 1.0)[:, 0].numpy()


# ------------------------------------------------------------------------------------------------ synthetic weights
def make_synthetic_weights(cfg: Qwen3TTSConfig, seed: int = 515) -> dict:
    """Talker-side weights, bf16 tensors, keys after sanitize.  LM stacks reuse oracle.llama's generator (mis-synth-v1)."""
    W = {}
    tw = llama_synth(cfg.talker, seed=seed)
    for k, v in tw.items():
        if k == "model.embed_tokens.weight":
            W["model.codec_embedding.weight"] = v
        elif k == "lm_head.weight":
            W["codec_head.weight"] = v
        else:
            W[k] = v
    pw = llama_synth(cfg.predictor, seed=seed + 1)
    for k, v in pw.items():
        if k.startswith("model.layers.") or k == "model.norm.weight":
            W["code_predictor." + k] = v
    key = [seed * 100000 + 50000]

    def mat(shape, amp):
        key[0] += 1
        return torch.from_numpy(synth.synth_tensor(key[0], shape, amp)).to(torch.bfloat16)
    d, dp, th = cfg.talker.hidden_size, cfg.predictor.hidden_size, cfg.text_hidden_size
    W["model.text_embedding.weight"] = mat((cfg.text_vocab_size, th), 0.5 * math.sqrt(3.0))
    W["text_projection.linear_fc1.weight"] = mat((th, th), math.sqrt(3.0 / th) * 2.0)
    W["text_projection.linear_fc1.bias"] = mat((th,), 0.1)
    W["text_projection.linear_fc2.weight"] = mat((d, th), math.sqrt(3.0 / th) * 2.0)
    W["text_projection.linear_fc2.bias"] = mat((d,), 0.1)
    for i in range(cfg.num_code_groups - 1):
        W[f"code_predictor.model.codec_embedding.{i}.weight"] = mat((cfg.predictor.vocab_size, d), 0.5 * math.sqrt(3.0))
        W[f"code_predictor.lm_head.{i}.weight"] = mat((cfg.predictor.vocab_size, dp), math.sqrt(3.0 / dp) * 2.0)
    if d != dp:
        W["code_predictor.small_to_mtp_projection.weight"] = mat((dp, d), math.sqrt(3.0 / d))
        W["code_predictor.small_to_mtp_projection.bias"] = mat((dp,), 0.1)
    return W


def make_synthetic_decoder_weights(cfg: DecoderConfig, seed: int = 616) -> dict:
    """float32, module-tree key names (what Qwen3TTSSpeechTokenizer.sanitize produces)."""
    W, key = {}, [seed * 100000]

    def t(shape, amp, offset=0.0):
        key[0] += 1
        return (synth.synth_tensor(key[0], shape, amp) + F(offset)).astype(F)

    half = cfg.codebook_dim // 2
    for name, n in (("rvq_first", cfg.num_semantic_quantizers), ("rvq_rest", cfg.num_quantizers - cfg.num_semantic_quantizers)):
        for i in range(n):
            p = f"decoder.quantizer.{name}.vq.layers.{i}.codebook"
            W[p + ".cluster_usage"] = t((cfg.codebook_size,), 0.5, 1.0)
            W[p + ".embedding_sum"] = t((cfg.codebook_size, half), math.sqrt(3.0) / math.sqrt(n))
        W[f"decoder.quantizer.{name}.output_proj.weight"] = t((cfg.codebook_dim, 1, half), math.sqrt(3.0 / half))

    def conv(p, co, k, ci, gain=1.0):
        W[p + ".weight"] = t((co, k, ci), gain * math.sqrt(3.0 / (k * ci)))
        W[p + ".bias"] = t((co,), 0.05)

    def lin(p, co, ci, bias=True, gain=1.0):
        W[p + ".weight"] = t((co, ci), gain * math.sqrt(3.0 / ci))
        if bias:
            W[p + ".bias"] = t((co,), 0.05)

    conv("decoder.pre_conv.conv", cfg.latent_dim, 3, cfg.codebook_dim)
    P, hs = "decoder.pre_transformer", cfg.hidden_size
    lin(P + ".input_proj", hs, cfg.latent_dim)
    lin(P + ".output_proj", cfg.latent_dim, hs)
    W[P + ".norm.weight"] = t((hs,), 0.1, 1.0)
    for i in range(cfg.num_hidden_layers):
        p = f"{P}.layers.{i}"
        W[p + ".input_layernorm.weight"] = t((hs,), 0.1, 1.0)
        W[p + ".post_attention_layernorm.weight"] = t((hs,), 0.1, 1.0)
        lin(p + ".self_attn.q_proj", cfg.num_attention_heads * cfg.head_dim, hs, bias=False, gain=1.5)
        lin(p + ".self_attn.k_proj", cfg.num_key_value_heads * cfg.head_dim, hs, bias=False, gain=1.5)
        lin(p + ".self_attn.v_proj", cfg.num_key_value_heads * cfg.head_dim, hs, bias=False)
        lin(p + ".self_attn.o_proj", hs, cfg.num_attention_heads * cfg.head_dim, bias=False)
        lin(p + ".mlp.gate_proj", cfg.intermediate_size, hs, bias=False)
        lin(p + ".mlp.up_proj", cfg.intermediate_size, hs, bias=False)
        lin(p + ".mlp.down_proj", hs, cfg.intermediate_size, bias=False)
        W[p + ".self_attn_layer_scale.scale"] = t((hs,), 0.2, 0.5)
        W[p + ".mlp_layer_scale.scale"] = t((hs,), 0.2, 0.5)
    ld = cfg.latent_dim
    for i, f in enumerate(cfg.upsampling_ratios):
        p = f"decoder.upsample.{i}.layers"
        conv(p + ".0.conv", ld, f, ld, gain=math.sqrt(f))                   # each output sample sees 1 of the f taps
        conv(p + ".1.dwconv.conv", ld, 7, 1)
        W[p + ".1.norm.weight"] = t((ld,), 0.1, 1.0); W[p + ".1.norm.bias"] = t((ld,), 0.05)
        lin(p + ".1.pwconv1", 4 * ld, ld)
        lin(p + ".1.pwconv2", ld, 4 * ld)
        W[p + ".1.gamma"] = t((ld,), 0.2, 0.4)
    conv("decoder.decoder.0.conv", cfg.decoder_dim, 7, ld)
    for bi, rate in enumerate(cfg.upsample_rates):
        cin, cout = cfg.decoder_dim >> bi, cfg.decoder_dim >> (bi + 1)
        p = f"decoder.decoder.{bi + 1}.block"
        W[p + ".0.alpha"] = t((cin,), 0.5); W[p + ".0.beta"] = t((cin,), 0.5)
        conv(p + ".1.conv", cout, 2 * rate, cin, gain=math.sqrt(rate) * 0.8)  # 2 of the 2*rate taps per output sample
        for ri in range(3):
            q = f"{p}.{ri + 2}"
            W[q + ".act1.alpha"] = t((cout,), 0.5); W[q + ".act1.beta"] = t((cout,), 0.5)
            conv(q + ".conv1.conv", cout, 7, cout, gain=0.7)
            W[q + ".act2.alpha"] = t((cout,), 0.5); W[q + ".act2.beta"] = t((cout,), 0.5)
            conv(q + ".conv2.conv", cout, 1, cout, gain=0.3)
    n, cl = len(cfg.upsample_rates), cfg.decoder_dim >> len(cfg.upsample_rates)
    W[f"decoder.decoder.{n + 1}.alpha"] = t((cl,), 0.5); W[f"decoder.decoder.{n + 1}.beta"] = t((cl,), 0.5)
    conv(f"decoder.decoder.{n + 2}.conv", 1, 7, cl, gain=0.5)
    return W
