"""Host mirror of Qwen3TTSModel : SpeechGenerationModel (Sources/MLXAudioTTS/Models/Qwen3TTS/Qwen3TTS.swift:12-133).

Device work (talker + code-predictor frame loop, sampleToken, speech-tokenizer decoder) is behind mis_qwen3tts_* in
libmi_speech.so.  Host logic mirrored here: the ChatML prompt template and the (text id, codec id) layout of
prepareGenerationInputs (Qwen3TTS.swift:883-1000), the per-utterance frame cap (:383) and the stream contract."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .codecs import _tensor_args
from .generation import (AudioEvent, AudioGenerationError, InfoEvent, TokenEvent, check, decode_audio_event,  # noqa: F401
                         stream_events)
from .tts import LlamaTTSConfiguration


def _lm(hidden, layers, ff, heads, kv, hd, vocab):
    return LlamaTTSConfiguration(hidden_size=hidden, num_hidden_layers=layers, intermediate_size=ff, num_attention_heads=heads,
                                 num_key_value_heads=kv, head_dim=hd, vocab_size=vocab, rms_norm_eps=1e-6, rope_theta=1e6,
                                 rope_scaling=None, tie_word_embeddings=False, qk_norm=True, rope_plain=True, rope_ops_in_dtype=True)


@dataclass
class Qwen3TTSDecoderConfiguration:
    """Qwen3TTSTokenizerDecoderConfig (Qwen3TTSConfig.swift:307-385)"""
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


@dataclass
class Qwen3TTSSpeakerEncoderConfiguration:
    """Qwen3TTSSpeakerEncoderConfig (Qwen3TTSConfig.swift:69-117)"""
    mel_dim: int = 128
    enc_dim: int = 1024
    enc_channels: tuple = (512, 512, 512, 512, 1536)
    enc_kernel_sizes: tuple = (5, 3, 3, 3, 1)
    enc_dilations: tuple = (1, 2, 3, 4, 1)
    enc_attention_channels: int = 128
    enc_res2net_scale: int = 8
    enc_se_channels: int = 128
    sample_rate: int = 24000


@dataclass
class Qwen3TTSTokenizerEncoderConfiguration:
    """Qwen3TTSTokenizerEncoderConfig (Qwen3TTSConfig.swift:388-497), Mimi defaults"""
    audio_channels: int = 1
    num_filters: int = 64
    kernel_size: int = 7
    last_kernel_size: int = 3
    residual_kernel_size: int = 3
    num_residual_layers: int = 1
    dilation_growth_rate: int = 2
    compress: int = 2
    upsampling_ratios: tuple = (8, 6, 5, 4)
    use_causal_conv: bool = True
    use_conv_shortcut: bool = False
    hidden_size: int = 512
    num_hidden_layers: int = 8
    num_attention_heads: int = 8
    intermediate_size: int = 2048
    rope_theta: float = 10000.0
    norm_eps: float = 1e-5
    sampling_rate: int = 24000
    frame_rate: float = 12.5
    codebook_dim: int = 256
    codebook_size: int = 2048
    num_quantizers: int = 32


def _dc_from(cls, d: dict):
    return cls(**{k: (tuple(d[k]) if isinstance(d[k], list) else d[k]) for k in cls.__dataclass_fields__ if k in d})


@dataclass
class Qwen3TTSConfiguration:
    """Qwen3TTSModelConfig + Qwen3TTSTalkerConfig (Qwen3TTSConfig.swift:200-305,532-583)"""
    talker: LlamaTTSConfiguration = field(default_factory=lambda: _lm(1024, 28, 3072, 16, 8, 128, 3072))
    predictor: LlamaTTSConfiguration = field(default_factory=lambda: _lm(1024, 5, 3072, 16, 8, 128, 2048))
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
    codec_language_id: dict | None = None
    spk_id: dict | None = None              # CustomVoice: speaker name -> codec token id (int or [int, ...]: first), Qwen3TTSConfig.swift:118-147
    spk_is_dialect: dict | None = None      # speaker name -> false | dialect name (a codec_language_id key), :153-196
    tts_model_type: str = "base"            # "custom_voice": `voice` = "speaker[, instruction]" (Qwen3TTS.swift:361-371)
    tts_pad_token_id: int = 151671
    tts_bos_token_id: int = 151672
    tts_eos_token_id: int = 151673
    sample_rate: int = 24000
    decoder: Qwen3TTSDecoderConfiguration = field(default_factory=Qwen3TTSDecoderConfiguration)
    # in-context voice cloning front end: None = not part of this checkpoint (no speaker encoder unless tts_model_type == "base",
    # Qwen3TTS.swift:46-48; no tokenizer encoder without encoder_config, Qwen3TTSSpeechTokenizer.swift:1041-1046)
    speaker_encoder: Qwen3TTSSpeakerEncoderConfiguration | None = None
    tokenizer_encoder: Qwen3TTSTokenizerEncoderConfiguration | None = None
    encoder_valid_num_quantizers: int = 16

    @classmethod
    def from_dict(cls, d: dict, tokenizer: dict | None = None) -> "Qwen3TTSConfiguration":
        """Qwen3TTSModelConfig (config.json: talker_config { code_predictor_config }, tts_*_token_id, sample_rate) plus the speech
        tokenizer's decoder_config (speech_tokenizer/config.json); defaults as Qwen3TTSConfig.swift:47-63,267-295,362-385,582-587."""
        t = d.get("talker_config") or {}
        cp = t.get("code_predictor_config") or {}

        def lm(c, layers, vocab):
            return _lm(c.get("hidden_size", 1024), c.get("num_hidden_layers", layers), c.get("intermediate_size", 3072),
                       c.get("num_attention_heads", 16), c.get("num_key_value_heads", 8), c.get("head_dim", 128), c.get("vocab_size", vocab))
        talker, pred = lm(t, 28, 3072), lm(cp, 5, 2048)
        talker.rms_norm_eps = t.get("rms_norm_eps", 1e-6); talker.rope_theta = t.get("rope_theta", 1e6)
        pred.rms_norm_eps = cp.get("rms_norm_eps", 1e-6); pred.rope_theta = cp.get("rope_theta", 1e6)
        dc = (tokenizer or {}).get("decoder_config") or {}
        dec = _dc_from(Qwen3TTSDecoderConfiguration, dc)
        spk = _dc_from(Qwen3TTSSpeakerEncoderConfiguration, d.get("speaker_encoder_config") or {}) if d.get("tts_model_type", "base") == "base" else None
        ec = (tokenizer or {}).get("encoder_config")
        enc = _dc_from(Qwen3TTSTokenizerEncoderConfiguration, ec) if ec is not None else None
        return cls(talker=talker, predictor=pred, num_code_groups=t.get("num_code_groups", 16),
                   text_hidden_size=t.get("text_hidden_size", 2048), text_vocab_size=t.get("text_vocab_size", 151936),
                   codec_eos_token_id=t.get("codec_eos_token_id", 2150), codec_think_id=t.get("codec_think_id", 2154),
                   codec_nothink_id=t.get("codec_nothink_id", 2155), codec_think_bos_id=t.get("codec_think_bos_id", 2156),
                   codec_think_eos_id=t.get("codec_think_eos_id", 2157), codec_pad_id=t.get("codec_pad_id", 2148),
                   codec_bos_id=t.get("codec_bos_id", 2149), codec_language_id=t.get("codec_language_id"),
                   spk_id=t.get("spk_id"), spk_is_dialect=t.get("spk_is_dialect"), tts_model_type=d.get("tts_model_type", "base"),
                   tts_pad_token_id=d.get("tts_pad_token_id", 151671), tts_bos_token_id=d.get("tts_bos_token_id", 151672),
                   tts_eos_token_id=d.get("tts_eos_token_id", 151673), sample_rate=d.get("sample_rate", 24000), decoder=dec,
                   speaker_encoder=spk, tokenizer_encoder=enc,
                   encoder_valid_num_quantizers=(tokenizer or {}).get("encoder_valid_num_quantizers", 16))

    def to_c(self) -> "_lib.Qwen3TTSConfigC":
        d = self.decoder
        ur = (C.c_int32 * 8)(*d.upsample_rates)
        up = (C.c_int32 * 8)(*d.upsampling_ratios)
        return _lib.Qwen3TTSConfigC(self.talker.to_c(), self.predictor.to_c(), self.num_code_groups, self.text_hidden_size,
                                    self.text_vocab_size, self.codec_eos_token_id, self.tts_pad_token_id, d.latent_dim,
                                    d.codebook_dim, d.codebook_size, d.decoder_dim, d.hidden_size, d.intermediate_size, d.head_dim,
                                    d.num_attention_heads, d.num_hidden_layers, d.num_key_value_heads, d.num_quantizers,
                                    d.num_semantic_quantizers, d.rms_norm_eps, d.rope_theta, len(d.upsample_rates), ur,
                                    len(d.upsampling_ratios), up, self.sample_rate)


    def reference_to_c(self) -> "_lib.Qwen3TTSReferenceConfigC":
        r = _lib.Qwen3TTSReferenceConfigC()
        sp, en = self.speaker_encoder, self.tokenizer_encoder
        if sp is not None:
            r.spk_mel_dim, r.spk_enc_dim, r.spk_n_blocks = sp.mel_dim, sp.enc_dim, len(sp.enc_channels)
            r.spk_channels = (C.c_int32 * 8)(*sp.enc_channels)
            r.spk_kernel_sizes = (C.c_int32 * 8)(*sp.enc_kernel_sizes)
            r.spk_dilations = (C.c_int32 * 8)(*sp.enc_dilations)
            r.spk_attention_channels, r.spk_res2net_scale = sp.enc_attention_channels, sp.enc_res2net_scale
            r.spk_se_channels, r.spk_sample_rate = sp.enc_se_channels, sp.sample_rate
        if en is not None:
            r.enc_audio_channels, r.enc_num_filters, r.enc_kernel_size = en.audio_channels, en.num_filters, en.kernel_size
            r.enc_last_kernel_size, r.enc_residual_kernel_size = en.last_kernel_size, en.residual_kernel_size
            r.enc_num_residual_layers, r.enc_dilation_growth_rate, r.enc_compress = en.num_residual_layers, en.dilation_growth_rate, en.compress
            r.enc_n_ratios = len(en.upsampling_ratios)
            r.enc_upsampling_ratios = (C.c_int32 * 8)(*en.upsampling_ratios)
            r.enc_use_causal_conv, r.enc_use_conv_shortcut = int(en.use_causal_conv), int(en.use_conv_shortcut)
            r.enc_hidden_size, r.enc_num_layers, r.enc_num_heads = en.hidden_size, en.num_hidden_layers, en.num_attention_heads
            r.enc_intermediate_size, r.enc_codebook_dim, r.enc_codebook_size = en.intermediate_size, en.codebook_dim, en.codebook_size
            r.enc_num_quantizers, r.enc_valid_num_quantizers = en.num_quantizers, self.encoder_valid_num_quantizers
            r.enc_sampling_rate, r.enc_rope_theta, r.enc_frame_rate, r.enc_norm_eps = en.sampling_rate, en.rope_theta, en.frame_rate, en.norm_eps
        return r


@dataclass
class Qwen3TTSGenerateParameters:
    """defaultGenerationParameters / resolveVoiceDesignGenerationSettings (Qwen3TTS.swift:30-37,651-664)"""
    max_tokens: int = 4096
    temperature: float = 0.9
    top_p: float = 1.0
    top_k: int = 0
    repetition_penalty: float = 1.05
    min_p: float = 0.0
    seed: int = 0
    row_offset: int = 0

    def to_c(self):
        return _lib.Qwen3TTSParamsC(int(self.max_tokens), float(self.temperature), float(self.top_p), int(self.top_k),
                                    float(self.repetition_penalty), float(self.min_p), int(self.seed), int(self.row_offset))


def read_safetensors(path: str) -> dict:
    """name -> torch tensor (bf16 / f16 / f32 as stored) or numpy uint32 array (quantised words).  MLX writes U32 for packed weights."""
    import json
    import torch
    out = {}
    with open(path, "rb") as f:
        n = int.from_bytes(f.read(8), "little")
        hdr = json.loads(f.read(n))
        blob = f.read()
    kinds = {"BF16": torch.bfloat16, "F16": torch.float16, "F32": torch.float32}
    for k, e in hdr.items():
        if k == "__metadata__":
            continue
        a, b = e["data_offsets"]
        raw = np.frombuffer(blob, np.uint8, b - a, a)
        if e["dtype"] in kinds:
            out[k] = torch.frombuffer(bytearray(raw.tobytes()), dtype=kinds[e["dtype"]]).reshape(e["shape"]) if b > a else torch.zeros(e["shape"], dtype=kinds[e["dtype"]])
        elif e["dtype"] in ("U32", "I32"):
            out[k] = raw.view(np.uint32).reshape(e["shape"]).copy()
        elif e["dtype"] in ("I64", "BOOL", "U8"):
            continue                                                     # bookkeeping tensors (codebook `initialized`, ...)
        else:
            raise AudioGenerationError(3, f"unsupported safetensors dtype {e['dtype']} for {k}")
    return out


def _mlx_conv_shape(shape) -> bool:
    """checkArrayShapeQwen3 (Qwen3TTSSpeechTokenizer.swift:1445-1455): does a 3-D conv weight already look like MLX's [out, k, in]?"""
    if len(shape) != 3:
        return False
    _, d2, d3 = shape
    if d2 == 1:
        return d3 > 64
    if d3 == 1:
        return d2 <= 64
    return d2 < d3


def sanitize_speech_tokenizer(weights: dict) -> dict:
    """Qwen3TTSSpeechTokenizer.sanitize (:1093-1440), decoder side: strip the speech_tokenizer./decoder_model. prefixes, keep the
    decoder codebooks' cluster_usage / embedding_sum (`_codebook.` -> `.codebook.`), transpose PyTorch conv weights to MLX's layout
    when the shape heuristic says they are not already (transposed convs [in, out, k] -> [out, k, in], convs [out, in, k] ->
    [out, k, in]), rename upsample.X.Y -> upsample.X.layers.Y.  Encoder side (:1240-1370,1411-1427): the HF Mimi names
    encoder.encoder.layers.N / encoder.encoder_transformer.layers.N / encoder.downsample / encoder.quantizer.* become the module
    tree of Qwen3TTSSpeechTokenizerEncoder behind `encoder_model.` (conv weights always transposed to [out, k, in], q/k/v stacked
    into in_proj, codebooks kept as cluster_usage + embedding_sum).  Speaker-encoder keys are not tokenizer weights (:1220-1222)."""
    import re
    out = {}
    enc_qkv, enc_cb = {}, {}
    conv_map = {0: "encoder.init_conv1d", 3: "encoder.layers.0.downsample", 6: "encoder.layers.1.downsample",
                9: "encoder.layers.2.downsample", 12: "encoder.layers.3.downsample", 14: "encoder.final_conv1d"}
    res_layer, res_block = {1: 0, 4: 1, 7: 2, 10: 3}, {1: 0, 3: 1}
    tl_names = {"self_attn.out_proj.weight": "self_attn.out_proj.weight", "self_attn.o_proj.weight": "self_attn.out_proj.weight",
                "mlp.fc1.weight": "gating.linear1.weight", "mlp.fc2.weight": "gating.linear2.weight",
                "input_layernorm.weight": "norm1.weight", "input_layernorm.bias": "norm1.bias",
                "post_attention_layernorm.weight": "norm2.weight", "post_attention_layernorm.bias": "norm2.bias",
                "self_attn_layer_scale.scale": "layer_scale_1.scale", "mlp_layer_scale.scale": "layer_scale_2.scale"}

    def t3(v):
        return v.permute(0, 2, 1).contiguous() if len(v.shape) == 3 else v

    def enc_group(path):
        if "rvq_first." in path or "semantic_residual_vector_quantizer" in path:
            return "rvq_first"
        return "rvq_rest"

    def encoder_key(k, v):
        parts = k.split(".")
        if k.startswith("encoder.encoder.layers."):
            if len(parts) < 4 or not parts[3].isdigit():
                return
            n = int(parts[3])
            if ".block." in k:
                if n in res_layer and len(parts) > 5 and parts[5].isdigit() and int(parts[5]) in res_block:
                    suffix = ".".join(parts[6:])
                    out[f"encoder_model.encoder.layers.{res_layer[n]}.residuals.0.block.{res_block[int(parts[5])]}.conv.{suffix}"] = \
                        t3(v) if suffix.endswith("weight") else v
            elif n in conv_map:
                suffix = ".".join(parts[4:])
                out[f"encoder_model.{conv_map[n]}.conv.{suffix}"] = t3(v) if suffix.endswith("weight") else v
            return
        if k.startswith("encoder.encoder_transformer.layers.") or k.startswith("encoder.encoder_transformer.transformer.layers."):
            off = 4 if (len(parts) >= 5 and parts[2] == "transformer" and parts[3] == "layers") else 3
            if len(parts) <= off or not parts[off].isdigit():
                return
            li, suffix = int(parts[off]), ".".join(parts[off + 1:])
            for nm in ("q", "k", "v"):
                if f"self_attn.{nm}_proj.weight" in suffix:
                    enc_qkv.setdefault(li, {})[nm] = v
                    return
            if "self_attn.qkv.weight" in suffix and len(v.shape) == 2:
                if v.shape[0] % 3 == 0 and v.shape[0] > 0:
                    h = v.shape[0] // 3
                    enc_qkv.setdefault(li, {}).update(q=v[:h], k=v[h:2 * h], v=v[2 * h:])
                return
            for src, dst in tl_names.items():
                if src in suffix:
                    out[f"encoder_model.encoder_transformer.transformer.layers.{li}.{dst}"] = v
                    return
            return
        if k.startswith("encoder.downsample."):
            suffix = k[len("encoder.downsample."):]
            out["encoder_model.downsample.conv.conv." + suffix] = t3(v) if suffix.endswith("weight") else v
            return
        if k.startswith("encoder.quantizer."):
            rest = k[len("encoder.quantizer."):]
            if ".codebook.embed.weight" in rest or rest.endswith("codebook.embed"):
                return
            if "codebook.cluster_usage" in rest or "codebook.embed_sum" in rest or "codebook.embedding_sum" in rest:
                base = rest.rsplit(".codebook.", 1)[0]
                enc_cb.setdefault(base, {})["cluster_usage" if "cluster_usage" in rest else "embedding_sum"] = v
                return
            if "codebook.initialized" in rest:
                return
            for proj in ("input_proj", "output_proj"):
                if proj + ".weight" in rest:
                    out[f"encoder_model.quantizer.{enc_group(rest)}.{proj}.weight"] = t3(v)
            return

    for raw, v in weights.items():
        k = raw
        stripped = True
        while stripped:
            stripped = False
            for pre in ("speech_tokenizer.", "encoder_model.", "decoder_model."):
                if k.startswith(pre):
                    k = k[len(pre):]; stripped = True; break
        if not k or "speaker_encoder" in k.split("."):
            continue
        if k.startswith("encoder."):
            if not isinstance(v, np.ndarray):
                encoder_key(k, v)
            continue
        if "initialized" in k:
            continue
        if "_codebook.cluster_usage" in k or "_codebook.embedding_sum" in k:
            base, leaf = k.rsplit("._codebook.", 1)
            out[f"{base}.codebook.{leaf}"] = v
            continue
        shape = tuple(v.shape)
        is_tconv = ("upsample" in k and ".0.conv.weight" in k) or ("decoder.decoder" in k and "block.1.conv.weight" in k)
        if len(shape) == 3 and not _mlx_conv_shape(shape):
            if is_tconv:
                v = v.permute(1, 2, 0).contiguous()
            elif "conv.weight" in k or "_proj.weight" in k:
                v = v.permute(0, 2, 1).contiguous()
        k = re.sub(r"upsample\.(\d+)\.(\d+)", r"upsample.\1.layers.\2", k)
        out[k] = v
    import torch
    for li, qkv in enc_qkv.items():
        if all(n in qkv for n in ("q", "k", "v")):
            out[f"encoder_model.encoder_transformer.transformer.layers.{li}.self_attn.in_proj.weight"] = torch.cat([qkv["q"], qkv["k"], qkv["v"]], 0)
    for base, data in enc_cb.items():
        parts = base.split(".")
        if "cluster_usage" not in data or "embedding_sum" not in data or "layers" not in parts:
            continue
        i = parts.index("layers")
        if i + 1 >= len(parts) or not parts[i + 1].isdigit():
            continue
        pre = f"encoder_model.quantizer.{enc_group(base)}.vq.layers.{int(parts[i + 1])}.codebook"
        out[pre + ".cluster_usage"] = data["cluster_usage"]
        out[pre + ".embedding_sum"] = data["embedding_sum"]
    return out


def sanitize_speaker_encoder(weights: dict) -> dict:
    """Qwen3TTSSpeakerEncoder.sanitize (Qwen3TTSSpeakerEncoder.swift:324-354): everything behind a `speaker_encoder` path component,
    3-D weights transposed to [out, k, in] unless the shape heuristic says they already are; keys come back as `speaker_encoder.<rest>`."""
    out = {}
    for raw, v in weights.items():
        parts = raw.split(".")
        if "speaker_encoder" not in parts or isinstance(v, np.ndarray):
            continue
        rest = ".".join(parts[parts.index("speaker_encoder") + 1:])
        if not rest:
            continue
        if rest.endswith(".weight") and len(v.shape) == 3 and not _mlx_conv_shape(tuple(v.shape)):
            v = v.permute(0, 2, 1).contiguous()
        out["speaker_encoder." + rest] = v
    return out


@dataclass
class ReferenceAudioContext:
    """ReferenceAudioContext (Qwen3TTS.swift:268-300) as the handle sees it: the speaker vector, the reference codes [n_q, T] and the
    prompt rows they occupy behind the codec vocabulary (speaker_row = -1: no speaker vector)."""
    speaker_embedding: np.ndarray | None
    codes: np.ndarray
    speaker_row: int
    first_frame_row: int


@dataclass
class PreparedPrompt:
    """One utterance as prepareGenerationInputs lays it out: prefill positions as (text id | -1, codec id | -1) pairs, the
    trailing text ids added to the generated frames, and the number of text tokens (frame cap, Qwen3TTS.swift:381-383)."""
    text_ids: np.ndarray
    codec_ids: np.ndarray
    trailing_ids: np.ndarray
    target_token_count: int = 0
    reference: object = None            # ReferenceAudioContext of an in-context prompt (its rows live on the model handle)


class Qwen3TTSModel:
    def __init__(self, config: Qwen3TTSConfiguration, device: int = 0):
        self.configuration = config
        self.device = device
        self.tokenizer = None
        self._h = C.c_void_p()
        cc = config.to_c()
        check(_lib.lib().mis_qwen3tts_create(C.byref(cc), device, C.byref(self._h)))
        self._ref_cache = None              # cachedReferenceAudioContext (Qwen3TTS.swift:268-300): one entry, keyed by the array object
        self._ref_log = []                  # every context registered on the handle, in row order (replayed onto replicas)
        if config.speaker_encoder is not None or config.tokenizer_encoder is not None:
            rc = config.reference_to_c()
            check(_lib.lib().mis_qwen3tts_enable_reference(self._h, C.byref(rc)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().mis_qwen3tts_destroy(h)

    @classmethod
    def from_weights(cls, config, weights: dict, device: int = 0) -> "Qwen3TTSModel":
        m = cls(config, device)
        for k, v in weights.items():
            m.set_tensor(k, v)
        m.finalize()
        return m

    # -- fromModelDirectory / fromPretrained (Qwen3TTS.swift:1122-1275) -----------------------------------------------------------
    @classmethod
    def from_model_directory(cls, model_dir: str, device: int = 0) -> "Qwen3TTSModel":
        """config.json + *.safetensors (talker, `talker.` prefix stripped by sanitize :357-365; quantised paths = those with a
        `.scales` companion, group size / bits from `quantization` and its per-layer overrides :1157-1170) + speech_tokenizer/
        (config.json, *.safetensors through `sanitize_speech_tokenizer`).  The in-context voice-cloning front end is loaded when the
        checkpoint carries it: `speaker_encoder.*` of a base model (:1222-1236) and the tokenizer's `encoder_model.*` (encoder_config)."""
        import json
        import os
        with open(os.path.join(model_dir, "config.json")) as f:
            cj = json.load(f)
        st_dir = os.path.join(model_dir, "speech_tokenizer")
        tj = {}
        if os.path.isdir(st_dir) and os.path.exists(os.path.join(st_dir, "config.json")):
            with open(os.path.join(st_dir, "config.json")) as f:
                tj = json.load(f)
        weights = {}
        for fn in sorted(os.listdir(model_dir)):
            if fn.endswith(".safetensors"):
                weights.update(read_safetensors(os.path.join(model_dir, fn)))
        tw = {}
        if os.path.isdir(st_dir):
            for fn in sorted(os.listdir(st_dir)):
                if fn.endswith(".safetensors"):
                    tw.update(read_safetensors(os.path.join(st_dir, fn)))
        config = Qwen3TTSConfiguration.from_dict(cj, tj)
        speaker = sanitize_speaker_encoder(weights)
        tokw = sanitize_speech_tokenizer(tw)
        if not speaker:
            config.speaker_encoder = None
        if not any(k.startswith("encoder_model.") for k in tokw):
            config.tokenizer_encoder = None
        m = cls(config, device)
        talker = {k[len("talker."):]: v for k, v in weights.items() if k.startswith("talker.")}
        quant = cj.get("quantization") or cj.get("quantization_config") or {}
        for name, arr in talker.items():
            if name.endswith(".scales") or name.endswith(".biases"):
                continue
            base = name[: -len(".weight")] if name.endswith(".weight") else name
            if base + ".scales" in talker:
                per = quant.get("talker." + base) or quant.get(base) or {}
                gs, bits = int(per.get("group_size", quant.get("group_size", 64))), int(per.get("bits", quant.get("bits", 4)))
                m.set_quantized_tensor(name, np.asarray(arr).view(np.uint32), talker[base + ".scales"], talker[base + ".biases"], gs, bits)
            else:
                m.set_tensor(name, arr)
        if not os.path.isdir(st_dir):
            raise AudioGenerationError(1, "speech_tokenizer directory not found: speech decoding unavailable")
        for name, arr in tokw.items():
            if name.startswith("decoder.") or (name.startswith("encoder_model.") and config.tokenizer_encoder is not None):
                m.set_tensor(name, arr)
        for name, arr in speaker.items():
            m.set_tensor(name, arr)
        m.finalize()
        return m

    @classmethod
    def from_pretrained(cls, model_repo: str, device: int = 0) -> "Qwen3TTSModel":
        import os
        if os.path.isdir(model_repo):
            return cls.from_model_directory(model_repo, device)
        raise AudioGenerationError(1, f"model repo {model_repo!r} is not a local directory (no network access)")

    def set_tensor(self, name: str, arr):
        keep, ptr, dt, shape = _tensor_args(arr)
        sh = (C.c_int64 * len(shape))(*shape)
        check(_lib.lib().mis_qwen3tts_set_tensor(self._h, name.encode(), ptr, dt, sh, len(shape)))

    def set_quantized_tensor(self, name: str, wq, scales, biases, group_size: int = 64, bits: int = 8):
        """A tensor of a quantised checkpoint: wq uint32 [N, K*bits/32], scales / biases [N, K/group_size]."""
        wq = np.ascontiguousarray(wq, dtype=np.uint32)
        ks, ps, ds, ss = _tensor_args(scales)
        kb, pb, db, sb = _tensor_args(biases)
        if ds != db or tuple(ss) != tuple(sb):
            raise AudioGenerationError(3, "scales and biases must share dtype and shape")
        N, K = int(ss[0]), int(ss[1]) * group_size
        check(_lib.lib().mis_qwen3tts_set_tensor_quantized(self._h, name.encode(), wq.ctypes.data, ps, pb, ds, N, K, group_size, bits))

    def finalize(self):
        check(_lib.lib().mis_qwen3tts_finalize(self._h))

    # -- protocol surface ------------------------------------------------------------------------------
    @property
    def sample_rate(self) -> int:
        return self.configuration.sample_rate

    @property
    def default_generation_parameters(self) -> Qwen3TTSGenerateParameters:
        return Qwen3TTSGenerateParameters()

    @property
    def samples_per_frame(self) -> int:
        return int(_lib.lib().mis_qwen3tts_samples_per_frame(self._h))

    @staticmethod
    def parse_custom_voice_prompt(voice: str | None):
        """parseCustomVoicePrompt (Qwen3TTS.swift:571-594): "speaker[, instruction]" -> (speaker, instruction | None); None if empty."""
        v = (voice or "").strip()
        if not v:
            return None
        if "," not in v:
            return v, None
        speaker, instruction = v.split(",", 1)
        speaker, instruction = speaker.strip(), instruction.strip()
        if not speaker:
            return v, None
        return speaker, (instruction or None)

    def prepare_generation_inputs(self, text: str, language: str = "auto", instruct: str | None = None,
                                  speaker: str | None = None) -> PreparedPrompt:
        """prepareGenerationInputs (Qwen3TTS.swift:883-1000).  The CustomVoice speaker (:914-936,957-962) is the talker's input
        embedding of the speaker's codec token, spliced between the think prefix and (pad, bos): on this side one more codec id, the
        engine embeds it with the same table; a dialect speaker overrides the language id."""
        if self.tokenizer is None:
            raise AudioGenerationError(1, "Qwen3TTS requires the text tokenizer to be loaded")
        cfg = self.configuration
        ids = list(self.tokenizer.encode(f"<|im_start|>assistant\n{text}<|im_end|>\n<|im_start|>assistant\n"))
        lang = None
        if language.lower() != "auto" and cfg.codec_language_id:
            lang = cfg.codec_language_id.get(language.lower())
        spk_token = None
        if speaker:
            v = (cfg.spk_id or {}).get(speaker.lower())
            if v is not None:                                           # SpkIdValue.intValue: the int, or the first of a list (0 if empty)
                spk_token = (int(v[0]) if v else 0) if isinstance(v, (list, tuple)) else int(v)
            dv = (cfg.spk_is_dialect or {}).get(speaker.lower())
            if isinstance(dv, str) and cfg.codec_language_id and dv in cfg.codec_language_id:
                lang = cfg.codec_language_id[dv]                          # dialect override (:927-935)
        prefix = ([cfg.codec_think_id, cfg.codec_think_bos_id, lang, cfg.codec_think_eos_id] if lang is not None
                  else [cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id])
        codec = prefix + ([spk_token] if spk_token is not None else []) + [cfg.codec_pad_id, cfg.codec_bos_id]
        t, c = [], []
        if instruct:
            ins = list(self.tokenizer.encode(f"<|im_start|>user\n{instruct}<|im_end|>\n"))
            t += ins; c += [-1] * len(ins)
        t += ids[:3]; c += [-1, -1, -1]                                  # role: <|im_start|>assistant\n
        pad_count = len(codec) - 2
        t += [cfg.tts_pad_token_id] * pad_count + [cfg.tts_bos_token_id]  # (pad..., bos) + codec[:-1]
        c += codec[:-1]
        t += [ids[3]]; c += [codec[-1]]                                   # first text token + codec_bos
        trailing = ids[4:len(ids) - 5] + [cfg.tts_eos_token_id]
        return PreparedPrompt(np.asarray(t, np.int32), np.asarray(c, np.int32), np.asarray(trailing, np.int32),
                              len(self.tokenizer.encode(text)))

    # -- in-context voice cloning (Qwen3TTS.swift:232-300,596-881) ------------------------------------------------------------
    @property
    def has_speaker_encoder(self) -> bool:
        return self.configuration.speaker_encoder is not None

    @property
    def has_tokenizer_encoder(self) -> bool:                      # speechTokenizer.hasEncoder
        return self.configuration.tokenizer_encoder is not None

    @staticmethod
    def _mono(ref_audio) -> np.ndarray:
        """referenceAudioForEncoder / extractSpeakerEmbedding shapes (:240-247,:842-862): [n], [1, n] or [1, 1, n] -> the first row"""
        a = np.asarray(ref_audio, np.float32)
        while a.ndim > 1:
            a = a[0]
        return np.ascontiguousarray(a)

    def extract_speaker_embedding(self, ref_audio) -> np.ndarray | None:
        """extractSpeakerEmbedding (:839-881): log-mel (24 kHz, nFft 1024, hop 256, 128 mels) -> ECAPA-TDNN, on the device."""
        if not self.has_speaker_encoder:
            return None
        a = self._mono(ref_audio)
        out = np.zeros(self.configuration.speaker_encoder.enc_dim, np.float32)
        check(_lib.lib().mis_qwen3tts_speaker_embedding(self._h, a.ctypes.data, len(a), out.ctypes.data))
        return out

    def encode_audio(self, ref_audio) -> np.ndarray:
        """speechTokenizer.encode (Qwen3TTSSpeechTokenizer.swift:1052-1058): waveform -> codes int32 [valid_num_quantizers, T]."""
        if not self.has_tokenizer_encoder:
            raise AudioGenerationError(1, "Encoder not available for this speech tokenizer")
        a = self._mono(ref_audio)
        out = C.c_void_p(); nq = C.c_int32(); nf = C.c_int32()
        check(_lib.lib().mis_qwen3tts_encode_audio(self._h, a.ctypes.data, len(a), C.byref(out), C.byref(nq), C.byref(nf)))
        try:
            return np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(nq.value, max(nf.value, 1)))[:, : nf.value].copy()
        finally:
            _lib.lib().mis_free(out)

    def reference_tap(self, kind: int, ref_audio, stage: int) -> np.ndarray:
        """parity tap of the front end: kind 0 speaker encoder, 1 tokenizer encoder -> [channels, length]"""
        a = self._mono(ref_audio)
        cfg = self.configuration
        if kind == 0:
            cap = 3 * max(cfg.speaker_encoder.enc_channels) * (len(a) // 256 + 8)
        else:
            cap = max(cfg.tokenizer_encoder.hidden_size * (len(a) // 2 + 8), 1 << 16)
        buf = np.zeros(cap, np.float32)
        ch = C.c_int32(); ln = C.c_int64()
        check(_lib.lib().mis_qwen3tts_reference_tap(self._h, kind, a.ctypes.data, len(a), stage, buf.ctypes.data, cap, C.byref(ch), C.byref(ln)))
        return buf[: ch.value * ln.value].reshape(ch.value, ln.value).copy()

    def add_reference(self, codes, speaker_embedding=None) -> "ReferenceAudioContext":
        """Registers reference codes [n_q, T] (+ speaker vector) on the handle: the rows prefill positions address (see
        mis_qwen3tts_add_reference).  Qwen3TTSReferenceConditioning with precomputed tensors comes in through here as well."""
        cd = np.ascontiguousarray(codes, np.int32)
        sv = None if speaker_embedding is None else np.ascontiguousarray(speaker_embedding, np.float32).reshape(-1)
        srow = C.c_int32(); frow = C.c_int32()
        check(_lib.lib().mis_qwen3tts_add_reference(self._h, cd.ctypes.data, cd.shape[0], cd.shape[1], None if sv is None else sv.ctypes.data,
                                                    0 if sv is None else len(sv), C.byref(srow), C.byref(frow)))
        self._ref_log.append((cd, sv))
        return ReferenceAudioContext(sv, cd, srow.value, frow.value)

    def clear_references(self):
        check(_lib.lib().mis_qwen3tts_clear_references(self._h))
        self._ref_log, self._ref_cache = [], None

    def reference_audio_context(self, ref_audio) -> "ReferenceAudioContext":
        """referenceAudioContext (:268-300): speaker embedding + reference codes + their ICL rows, cached for the same array object."""
        if self._ref_cache is not None and self._ref_cache[0] is ref_audio:
            return self._ref_cache[1]
        ctx = self.add_reference(self.encode_audio(ref_audio), self.extract_speaker_embedding(ref_audio))
        self._ref_cache = (ref_audio, ctx)
        return ctx

    def prepare_reference_conditioning(self, ref_audio, ref_text: str, language: str | None = None, speaker_embedding=None):
        """prepareReferenceConditioning (:704-751): (context, reference text ids, codec language id | None).  A caller-supplied
        speaker vector replaces the extracted one (registered as its own context)."""
        if self.tokenizer is None:
            raise AudioGenerationError(1, "Qwen3TTS reference conditioning requires the text tokenizer to be loaded")
        if not self.has_tokenizer_encoder:
            raise AudioGenerationError(3, "Qwen3TTS reference conditioning requires a speech tokenizer encoder, but this checkpoint does not provide one.")
        ctx = self.reference_audio_context(ref_audio)
        if speaker_embedding is not None:
            ctx = self.add_reference(ctx.codes, speaker_embedding)
        lang = (language or "auto").lower()
        ids = list(self.tokenizer.encode(f"<|im_start|>assistant\n{ref_text}<|im_end|>\n"))
        start = min(3, len(ids))
        ref_ids = ids[start:max(start, len(ids) - 2)]
        lid = None
        if lang != "auto" and self.configuration.codec_language_id:
            lid = self.configuration.codec_language_id.get(lang)
        return ctx, ref_ids, lid

    def prepare_icl_generation_inputs(self, text: str, ref_audio=None, ref_text: str | None = None, language: str = "auto",
                                      conditioning=None) -> PreparedPrompt:
        """prepareICLGenerationInputs (:753-837) as (text id, codec id) positions: role, the think prefix with the speaker row,
        [reference text + target text + tts_eos] over codec_pad, [codec_bos + the reference frames' code-sum rows] over tts_pad.
        Everything is prefilled; the generated frames get tts_pad (trailingTextHidden = ttsPadEmbed)."""
        if self.tokenizer is None:
            raise AudioGenerationError(1, "Qwen3TTS request assembly requires the text tokenizer to be loaded")
        cfg = self.configuration
        ctx, ref_ids, lid = conditioning if conditioning is not None else self.prepare_reference_conditioning(ref_audio, ref_text, language)
        V = cfg.talker.vocab_size
        ids = list(self.tokenizer.encode(f"<|im_start|>assistant\n{text}<|im_end|>\n<|im_start|>assistant\n"))
        start = min(3, len(ids))
        target = ids[start:max(start, len(ids) - 5)]
        prefix = ([cfg.codec_think_id, cfg.codec_think_bos_id, lid, cfg.codec_think_eos_id] if lid is not None
                  else [cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id])
        codec = prefix + ([V + ctx.speaker_row] if ctx.speaker_row >= 0 else []) + [cfg.codec_pad_id, cfg.codec_bos_id]
        t = ids[:3]; c = [-1] * len(t)
        t += [cfg.tts_pad_token_id] * (len(codec) - 2) + [cfg.tts_bos_token_id]; c += codec[:-1]
        body = list(ref_ids) + target + [cfg.tts_eos_token_id]
        t += body; c += [cfg.codec_pad_id] * len(body)
        T = ctx.codes.shape[1]
        t += [cfg.tts_pad_token_id] * (T + 1); c += [cfg.codec_bos_id] + [V + ctx.first_frame_row + i for i in range(T)]
        return PreparedPrompt(np.asarray(t, np.int32), np.asarray(c, np.int32), np.zeros(0, np.int32), len(self.tokenizer.encode(text)), ctx)

    def _sync_references(self, replicas):
        """replicas decode their own rows: every replica needs the contexts of this model, at the same rows"""
        for r in replicas:
            if r is self:
                continue
            for cd, sv in self._ref_log[len(r._ref_log):]:
                r.add_reference(cd, sv)

    def _marshal(self, prompts):
        B = len(prompts)
        if B == 0:
            raise AudioGenerationError(3, "empty batch")
        P = max(len(p.text_ids) for p in prompts)
        Tt = max(max(len(p.trailing_ids) for p in prompts), 1)
        t = np.full((B, P), -1, np.int32); c = np.full((B, P), -1, np.int32); tr = np.zeros((B, Tt), np.int32)
        pl = np.zeros(B, np.int32); tl = np.zeros(B, np.int32)
        for b, p in enumerate(prompts):
            n = len(p.text_ids)
            t[b, :n] = p.text_ids; c[b, :n] = p.codec_ids; pl[b] = n
            tl[b] = len(p.trailing_ids); tr[b, :tl[b]] = p.trailing_ids
        return t, c, pl, P, tr, tl, Tt

    def _row_caps(self, prompts, gp):
        caps = np.asarray([min(gp.max_tokens, max(75, p.target_token_count * 6)) if p.target_token_count > 0 else gp.max_tokens
                           for p in prompts], np.int32)                  # effectiveMaxTokens (:383)
        return caps

    def generate_codes(self, prompts, generation_parameters: Qwen3TTSGenerateParameters | None = None):
        """Frame loop only: list of [n_frames, num_code_groups] int32 arrays."""
        gp = generation_parameters or self.default_generation_parameters
        t, c, pl, P, tr, tl, Tt = self._marshal(prompts)
        B = len(prompts)
        caps = self._row_caps(prompts, gp)
        gpc = gp.to_c()
        gpc.max_frames = int(caps.max())
        out = C.c_void_p(); stride = C.c_int64(); nf = (C.c_int32 * B)()
        check(_lib.lib().mis_qwen3tts_generate_codes(self._h, t.ctypes.data, c.ctypes.data, pl.ctypes.data, P, tr.ctypes.data,
                                                     tl.ctypes.data, Tt, B, C.byref(gpc), caps.ctypes.data, C.byref(out),
                                                     C.byref(stride), nf))
        try:
            G = self.configuration.num_code_groups
            arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(B, max(stride.value, 1), G))
            return [arr[b, : nf[b]].copy() for b in range(B)]
        finally:
            _lib.lib().mis_free(out)

    def decode_codes(self, codes) -> np.ndarray:
        """speechTokenizer.decoder over whole sequences: codes [B, num_quantizers, T] -> [B, T * samples_per_frame]."""
        cd = np.ascontiguousarray(codes, dtype=np.int32)
        B, nq, T = cd.shape
        out = np.zeros((B, T * self.samples_per_frame), np.float32)
        check(_lib.lib().mis_qwen3tts_decode(self._h, cd.ctypes.data, B, T, out.ctypes.data))
        return out

    def decoder_tap(self, codes, stage: int) -> np.ndarray:
        cd = np.ascontiguousarray(codes, dtype=np.int32)
        B, nq, T = cd.shape
        d = self.configuration.decoder
        cap = B * max(d.decoder_dim, d.latent_dim, d.codebook_dim) * T * self.samples_per_frame
        buf = np.zeros(cap, np.float32)
        ch = C.c_int32(); ln = C.c_int64()
        check(_lib.lib().mis_qwen3tts_decoder_tap(self._h, cd.ctypes.data, B, T, stage, buf.ctypes.data, cap, C.byref(ch), C.byref(ln)))
        return buf[: B * ch.value * ln.value].reshape(B, ch.value, ln.value).copy()

    def generate_batch(self, prompts, generation_parameters: Qwen3TTSGenerateParameters | None = None, return_codes: bool = False,
                       streaming_interval: float | None = None, on_audio=None, replicas=None):
        """generateVoiceDesign for a batch of prepared prompts: list of 1-D float32 PCM arrays.  With `on_audio` the decoded
        audio is also delivered in chunks of streaming_interval * 12.5 frames (Qwen3TTS.swift:394-395)."""
        gp = generation_parameters or self.default_generation_parameters
        t, c, pl, P, tr, tl, Tt = self._marshal(prompts)
        B = len(prompts)
        caps = self._row_caps(prompts, gp)
        gpc = gp.to_c()
        gpc.max_frames = int(caps.max())
        pcm = C.c_void_p(); stride = C.c_int64(); plens = (C.c_int64 * B)()
        codes = C.c_void_p(); cstride = C.c_int64(); nf = (C.c_int32 * B)()
        chunk = max(1, int((streaming_interval or 2.0) * 12.5))
        keep = []

        def cb(user, row, kind, payload, n):
            if kind == _lib.EVENT_AUDIO and on_audio is not None:
                on_audio(row, np.ctypeslib.as_array(C.cast(payload, C.POINTER(C.c_float)), shape=(n,)).copy())
        cbc = _lib.EVENT_CB(cb) if on_audio is not None else C.cast(None, _lib.EVENT_CB)
        keep.append(cbc)
        if replicas:          # Qwen3TTSModel objects with the same weights, one per GPU: rows sharded inside the library, each replica
            self._sync_references(replicas)
            hs = (C.c_void_p * len(replicas))(*[r._h for r in replicas])          # streams its own rows' chunks (global row indices)
            check(_lib.lib().mis_qwen3tts_group_generate(hs, len(replicas), t.ctypes.data, c.ctypes.data, pl.ctypes.data, P, tr.ctypes.data,
                                                         tl.ctypes.data, Tt, B, C.byref(gpc), caps.ctypes.data, C.byref(pcm), C.byref(stride),
                                                         plens, C.byref(codes), C.byref(cstride), nf, chunk, cbc, None, None))
        else:
            check(_lib.lib().mis_qwen3tts_generate(self._h, t.ctypes.data, c.ctypes.data, pl.ctypes.data, P, tr.ctypes.data, tl.ctypes.data,
                                                   Tt, B, C.byref(gpc), caps.ctypes.data, C.byref(pcm), C.byref(stride), plens,
                                                   C.byref(codes), C.byref(cstride), nf, chunk, cbc, None, None))
        try:
            arr = np.ctypeslib.as_array(C.cast(pcm, C.POINTER(C.c_float)), shape=(B, max(stride.value, 1)))
            out = [arr[b, : plens[b]].copy() for b in range(B)]
            G = self.configuration.num_code_groups
            ca = np.ctypeslib.as_array(C.cast(codes, C.POINTER(C.c_int32)), shape=(B, max(cstride.value, 1), G))
            cl = [ca[b, : nf[b]].copy() for b in range(B)]
        finally:
            _lib.lib().mis_free(pcm)
            _lib.lib().mis_free(codes)
        return (out, cl) if return_codes else out

    def generate(self, text: str, voice: str | None = None, ref_audio=None, ref_text=None, language: str | None = None,
                 generation_parameters: Qwen3TTSGenerateParameters | None = None) -> np.ndarray:
        """generate(text:voice:refAudio:refText:language:generationParameters:) (Qwen3TTS.swift:60-82); `voice` is the
        VoiceDesign instruction."""
        p = self._prepare(text, voice, language, ref_audio, ref_text)
        out = self.generate_batch([p], generation_parameters)[0]
        return out if len(out) else np.zeros(1, np.float32)              # generatedCodes.isEmpty -> zeros([1]) (:520-522)

    def _prepare(self, text, voice, language, ref_audio=None, ref_text=None):
        """the input branches of generateVoiceDesign (Qwen3TTS.swift:337-377): reference audio + text on a tokenizer with an encoder ->
        in-context prompt; otherwise CustomVoice models read `voice` as "speaker[, instruction]", the others as the VoiceDesign
        instruction"""
        if ref_audio is not None and ref_text is not None and self.has_tokenizer_encoder:
            return self.prepare_icl_generation_inputs(text, ref_audio, ref_text, language or "auto")
        if self.configuration.tts_model_type == "custom_voice":
            cv = self.parse_custom_voice_prompt(voice)
            return self.prepare_generation_inputs(text, language or "auto", cv[1] if cv else None, cv[0] if cv else None)
        return self.prepare_generation_inputs(text, language or "auto", voice)

    # -- streamingStep / resetStreamingState (Qwen3TTSSpeechTokenizer.swift:948-1006) --------------------------------------
    def set_stream_exact(self, exact: bool):
        """False (default): the reference's streaming arithmetic (bias counted twice after chunk boundaries, :556-559);
        True: chunked decode bitwise equal to decode_codes of the whole sequence."""
        check(_lib.lib().mis_qwen3tts_set_stream_exact(self._h, 1 if exact else 0))

    def reset_streaming_state(self, batch: int = 1, max_frames: int = 4096, max_chunk_frames: int = 64):
        check(_lib.lib().mis_qwen3tts_decode_stream_begin(self._h, batch, max_frames, max_chunk_frames))
        self._stream_batch = batch

    def streaming_step(self, codes) -> np.ndarray:
        """codes [B, num_quantizers, Tn] = only the new frames -> [B, Tn * samples_per_frame]; state stays on the device."""
        cd = np.ascontiguousarray(codes, dtype=np.int32)
        B, nq, T = cd.shape
        if B != getattr(self, "_stream_batch", None):
            raise AudioGenerationError(3, "streaming_step: batch differs from reset_streaming_state")
        out = np.zeros((B, T * self.samples_per_frame), np.float32)
        check(_lib.lib().mis_qwen3tts_decode_stream_step(self._h, cd.ctypes.data, T, out.ctypes.data))
        return out

    def end_streaming(self):
        check(_lib.lib().mis_qwen3tts_decode_stream_end(self._h))

    def generate_stream_batch(self, prompts, generation_parameters: Qwen3TTSGenerateParameters | None = None,
                              streaming_interval: float = 2.0, cancel_flag=None):
        """mis_qwen3tts_generate in streaming mode for a batch of prepared prompts: TokenEvent (code 0 of each frame) and
        AudioEvent chunks of streaming_interval * 12.5 frames WHILE the engine generates, InfoEvent per row when the frame loop
        ends, then the frames after the last full chunk."""
        gp = generation_parameters or self.default_generation_parameters
        t, c, pl, P, tr, tl, Tt = self._marshal(prompts)
        B = len(prompts)
        caps = self._row_caps(prompts, gp)
        gpc = gp.to_c()
        gpc.max_frames = int(caps.max())
        chunk = max(1, int(streaming_interval * 12.5))                  # streamingChunkSize (:394-395)
        pcm = C.c_void_p(); stride = C.c_int64(); plens = (C.c_int64 * B)()

        def start(cbf, flag_addr):
            st = _lib.lib().mis_qwen3tts_generate(self._h, t.ctypes.data, c.ctypes.data, pl.ctypes.data, P, tr.ctypes.data, tl.ctypes.data,
                                                  Tt, B, C.byref(gpc), caps.ctypes.data, C.byref(pcm), C.byref(stride), plens,
                                                  None, None, None, chunk, cbf, None, flag_addr)
            if pcm.value:
                _lib.lib().mis_free(pcm)
            return st
        yield from stream_events(start, decode_audio_event, cancel_flag)

    def generate_stream(self, text: str, voice: str | None = None, language: str | None = None,
                        generation_parameters: Qwen3TTSGenerateParameters | None = None, streaming_interval: float = 2.0,
                        ref_audio=None, ref_text=None):
        """generateStream (:84-133): .token per frame and .audio chunks while generating, .info when the loop ends, then the
        remaining samples."""
        p = self._prepare(text, voice, language, ref_audio, ref_text)
        yield from self.generate_stream_batch([p], generation_parameters, streaming_interval)
