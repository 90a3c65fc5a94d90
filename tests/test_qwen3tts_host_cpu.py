"""CPU: host-side prompt layout of the Qwen3-TTS mirror (prepareGenerationInputs, Qwen3TTS.swift:883-1000)."""
import numpy as np

from mlx_audio_swift_amd import qwen3tts as q3


class _Tok:
    def encode(self, s):
        out, i = [], 0
        special = {"<|im_start|>": 151644, "<|im_end|>": 151645, "assistant": 77091, "user": 872, "\n": 198}
        while i < len(s):
            for k, v in special.items():
                if s.startswith(k, i):
                    out.append(v); i += len(k); break
            else:
                out.append(1000 + ord(s[i])); i += 1
        return out


def test_prompt_layout_matches_reference_construction():
    cfg = q3.Qwen3TTSConfiguration(codec_language_id={"english": 2050})
    m = object.__new__(q3.Qwen3TTSModel)
    m.configuration = cfg; m.tokenizer = _Tok(); m._h = None
    p = m.prepare_generation_inputs("Hi you", "auto", None)
    ids = _Tok().encode("<|im_start|>assistant\nHi you<|im_end|>\n<|im_start|>assistant\n")
    # role (3 text-only) | pad pad pad+nothink.. | bos + think_eos | first text token + codec_bos
    assert p.text_ids.tolist() == ids[:3] + [cfg.tts_pad_token_id] * 3 + [cfg.tts_bos_token_id] + [ids[3]]
    assert p.codec_ids.tolist() == [-1, -1, -1, cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id,
                                    cfg.codec_pad_id, cfg.codec_bos_id]
    assert p.trailing_ids.tolist() == ids[4:len(ids) - 5] + [cfg.tts_eos_token_id] and p.target_token_count == 6
    q = m.prepare_generation_inputs("Hi you", "English", "calm voice")
    ins = _Tok().encode("<|im_start|>user\ncalm voice<|im_end|>\n")
    assert q.text_ids.tolist()[: len(ins)] == ins and q.codec_ids.tolist()[: len(ins)] == [-1] * len(ins)
    assert q.codec_ids.tolist()[len(ins) + 3:] == [cfg.codec_think_id, cfg.codec_think_bos_id, 2050, cfg.codec_think_eos_id,
                                                   cfg.codec_pad_id, cfg.codec_bos_id]
    assert len(q.text_ids) == len(q.codec_ids) == len(ins) + 3 + 6
    t, c, pl, P, tr, tl, Tt = m._marshal([p, q])
    assert t.shape == (2, P) and pl.tolist() == [len(p.text_ids), len(q.text_ids)] and tl.tolist() == [len(p.trailing_ids)] * 2
    assert m._row_caps([p, q], q3.Qwen3TTSGenerateParameters(max_tokens=4096)).tolist() == [75, 75]


def test_speech_tokenizer_sanitize_and_safetensors_reader_round_trip(tmp_path):
    """Qwen3TTSSpeechTokenizer.sanitize (Qwen3TTSSpeechTokenizer.swift:1093-1440), decoder side: a PyTorch-layout checkpoint
    ([out, in, k] convs, [in, out, k] transposed convs, `_codebook.` statistics, upsample.X.Y names, encoder keys) comes back as the
    module-tree tensors the engine takes; the U32 words of quantised tensors survive the reader."""
    import torch
    from safetensors.torch import save_file
    from mlx_audio_swift_amd.qwen3tts import read_safetensors, sanitize_speech_tokenizer, Qwen3TTSConfiguration
    from oracle import qwen3tts as oq
    dcfg = oq.DecoderConfig(**{**oq.TINY.decoder.__dict__, "codebook_dim": 160, "decoder_dim": 288})
    Wd = oq.make_synthetic_decoder_weights(dcfg)
    pt = {"encoder.encoder.layers.0.conv.weight": torch.zeros(4, 1, 7), "decoder.quantizer.rvq_first.vq.layers.0._codebook.initialized": torch.ones(1)}
    for k, v in Wd.items():
        v = torch.as_tensor(v)
        if ".codebook." in k:
            k = k.replace(".codebook.", "._codebook.")
        elif v.ndim == 3:
            tconv = ("upsample" in k and ".0.conv.weight" in k) or ("decoder.decoder" in k and "block.1.conv.weight" in k)
            v = v.permute(2, 0, 1) if tconv else v.permute(0, 2, 1)
        k = k.replace(".layers.0.", ".0.").replace(".layers.1.", ".1.") if k.startswith("decoder.upsample.") else k
        pt["speech_tokenizer." + k if "quantizer" in k else k] = v.contiguous()
    save_file(pt, str(tmp_path / "m.safetensors"))
    back = sanitize_speech_tokenizer(read_safetensors(str(tmp_path / "m.safetensors")))
    assert {k for k in back if not k.startswith("encoder_model.")} == set(Wd)
    assert tuple(back["encoder_model.encoder.init_conv1d.conv.conv.weight"].shape) == (4, 7, 1)       # encoder convs are always transposed
    for k, v in Wd.items():
        assert tuple(back[k].shape) == tuple(np.asarray(v).shape) and np.array_equal(back[k].float().numpy(), torch.as_tensor(v).float().numpy()), k
    cfg = Qwen3TTSConfiguration.from_dict({"talker_config": {"hidden_size": 256, "code_predictor_config": {"num_hidden_layers": 3}}, "tts_pad_token_id": 7},
                                          {"decoder_config": {"upsample_rates": [3, 2], "codebook_dim": 160}})
    assert (cfg.talker.hidden_size, cfg.talker.num_hidden_layers, cfg.predictor.num_hidden_layers, cfg.predictor.vocab_size) == (256, 28, 3, 2048)
    assert cfg.tts_pad_token_id == 7 and cfg.decoder.upsample_rates == (3, 2) and cfg.decoder.codebook_dim == 160 and cfg.decoder.latent_dim == 1024


def test_custom_voice_speaker_branch_and_dialect_override():
    """CustomVoice models (Qwen3TTS.swift:361-371,914-936,957-962): `voice` = "speaker[, instruction]"; the speaker's codec token is
    spliced between the think prefix and (pad, bos); a dialect speaker overrides the language id; unknown speakers fall through."""
    cfg = q3.Qwen3TTSConfiguration(codec_language_id={"english": 2050, "sichuan_dialect": 2062}, tts_model_type="custom_voice",
                                   spk_id={"ryan": 3061, "eric": [3065, 7]}, spk_is_dialect={"ryan": False, "eric": "sichuan_dialect"})
    m = object.__new__(q3.Qwen3TTSModel)
    m.configuration = cfg; m.tokenizer = _Tok(); m._h = None
    assert q3.Qwen3TTSModel.parse_custom_voice_prompt(" Ryan , speak slowly ") == ("Ryan", "speak slowly")
    assert q3.Qwen3TTSModel.parse_custom_voice_prompt("Ryan") == ("Ryan", None)
    assert q3.Qwen3TTSModel.parse_custom_voice_prompt(", x") == (", x", None) and q3.Qwen3TTSModel.parse_custom_voice_prompt("  ") is None
    p = m._prepare("Hi you", "Ryan", "English")
    tail = [cfg.codec_think_id, cfg.codec_think_bos_id, 2050, cfg.codec_think_eos_id, 3061, cfg.codec_pad_id, cfg.codec_bos_id]
    assert p.codec_ids.tolist() == [-1, -1, -1] + tail
    ids = _Tok().encode("<|im_start|>assistant\nHi you<|im_end|>\n<|im_start|>assistant\n")
    assert p.text_ids.tolist() == ids[:3] + [cfg.tts_pad_token_id] * (len(tail) - 2) + [cfg.tts_bos_token_id] + [ids[3]]   # padCount = prefix - 2
    q = m._prepare("Hi you", "eric, whisper", "auto")                        # dialect speaker: language id from spk_is_dialect
    ins = _Tok().encode("<|im_start|>user\nwhisper<|im_end|>\n")
    assert q.codec_ids.tolist()[len(ins) + 3:] == [cfg.codec_think_id, cfg.codec_think_bos_id, 2062, cfg.codec_think_eos_id, 3065,
                                                   cfg.codec_pad_id, cfg.codec_bos_id]
    r = m._prepare("Hi you", "nobody", "auto")                               # unknown speaker: no splice (the reference only warns)
    assert r.codec_ids.tolist() == [-1, -1, -1, cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id, cfg.codec_pad_id, cfg.codec_bos_id]
    base = q3.Qwen3TTSConfiguration.from_dict({"tts_model_type": "custom_voice", "talker_config": {"spk_id": {"a": 5}, "spk_is_dialect": {"a": False}}})
    assert base.tts_model_type == "custom_voice" and base.spk_id == {"a": 5} and base.spk_is_dialect == {"a": False}


def test_encoder_side_sanitize_and_speaker_encoder_sanitize():
    """Qwen3TTSSpeechTokenizer.sanitize, encoder side (:1099-1108,1240-1370,1411-1427): the HF Mimi names of a published speech tokenizer
    come back as the module tree of Qwen3TTSSpeechTokenizerEncoder behind `encoder_model.` - exactly the keys and tensors of the oracle's
    weight dict; Qwen3TTSSpeakerEncoder.sanitize (Qwen3TTSSpeakerEncoder.swift:324-354) for the x-vector net."""
    import torch
    from mlx_audio_swift_amd.qwen3tts import sanitize_speaker_encoder, sanitize_speech_tokenizer, Qwen3TTSConfiguration
    from oracle import ecapa as oe
    from oracle import mimi_encoder as om
    cfg = om.MimiEncoderConfig(num_filters=4, hidden_size=16, num_hidden_layers=2, num_attention_heads=2, intermediate_size=32,
                               codebook_dim=8, codebook_size=32, num_quantizers=5, valid_num_quantizers=4)
    W = {k: torch.as_tensor(v) for k, v in om.make_synthetic_weights(cfg).items()}
    hf = {}

    def conv(hf_name, o_name, bias=True):
        hf["encoder." + hf_name + ".conv.weight"] = W[o_name + ".conv.conv.weight"].permute(0, 2, 1).contiguous()       # PyTorch [out, in, k]
        if bias:
            hf["encoder." + hf_name + ".conv.bias"] = W[o_name + ".conv.conv.bias"]
    conv("encoder.layers.0", "encoder.init_conv1d")
    idx = 1
    for li in range(4):
        conv(f"encoder.layers.{idx}.block.1", f"encoder.layers.{li}.residuals.0.block.0")
        conv(f"encoder.layers.{idx}.block.3", f"encoder.layers.{li}.residuals.0.block.1")
        idx += 2
        conv(f"encoder.layers.{idx}", f"encoder.layers.{li}.downsample")
        idx += 1
    conv(f"encoder.layers.{idx + 1}", "encoder.final_conv1d")
    D = cfg.hidden_size
    for li in range(cfg.num_hidden_layers):
        p, q = f"encoder_transformer.transformer.layers.{li}", f"speech_tokenizer.encoder.encoder_transformer.layers.{li}"
        w = W[p + ".self_attn.in_proj.weight"]
        hf[q + ".self_attn.q_proj.weight"], hf[q + ".self_attn.k_proj.weight"], hf[q + ".self_attn.v_proj.weight"] = w[:D], w[D:2 * D], w[2 * D:]
        hf[q + ".self_attn.o_proj.weight"] = W[p + ".self_attn.out_proj.weight"]
        hf[q + ".mlp.fc1.weight"], hf[q + ".mlp.fc2.weight"] = W[p + ".gating.linear1.weight"], W[p + ".gating.linear2.weight"]
        hf[q + ".input_layernorm.weight"], hf[q + ".input_layernorm.bias"] = W[p + ".norm1.weight"], W[p + ".norm1.bias"]
        hf[q + ".post_attention_layernorm.weight"], hf[q + ".post_attention_layernorm.bias"] = W[p + ".norm2.weight"], W[p + ".norm2.bias"]
        hf[q + ".self_attn_layer_scale.scale"], hf[q + ".mlp_layer_scale.scale"] = W[p + ".layer_scale_1.scale"], W[p + ".layer_scale_2.scale"]
    hf["encoder.downsample.conv.weight"] = W["downsample.conv.conv.conv.weight"].permute(0, 2, 1).contiguous()
    for hf_grp, o_grp, nq in (("semantic", "rvq_first", 1), ("acoustic", "rvq_rest", cfg.num_quantizers - 1)):
        hp, op = f"encoder.quantizer.{hf_grp}_residual_vector_quantizer", f"quantizer.{o_grp}"
        hf[hp + ".input_proj.weight"] = W[op + ".input_proj.weight"].permute(0, 2, 1).contiguous()
        hf[hp + ".output_proj.weight"] = torch.zeros(cfg.hidden_size, cfg.codebook_dim, 1)
        for i in range(nq):
            hf[f"{hp}.layers.{i}.codebook.embed_sum"] = W[f"{op}.vq.layers.{i}.codebook.embedding_sum"]
            hf[f"{hp}.layers.{i}.codebook.cluster_usage"] = W[f"{op}.vq.layers.{i}.codebook.cluster_usage"]
            hf[f"{hp}.layers.{i}.codebook.initialized"] = torch.ones(1)
    hf["speaker_encoder.fc.weight"] = torch.zeros(3, 4, 1)                      # not a tokenizer weight (:1220-1222)
    back = sanitize_speech_tokenizer(hf)
    enc = {k[len("encoder_model."):]: v for k, v in back.items() if k.startswith("encoder_model.") and "output_proj" not in k}
    assert set(enc) == set(W), (set(enc) ^ set(W))
    for k, v in W.items():
        assert tuple(enc[k].shape) == tuple(v.shape) and torch.equal(enc[k], v), k
    assert "encoder_model.quantizer.rvq_first.output_proj.weight" in back and not any("speaker_encoder" in k for k in back)
    # speaker encoder: PyTorch conv layout [out, in, k] -> [out, k, in] when the heuristic says so; any prefix before `speaker_encoder`
    ecfg = oe.EcapaConfig()
    Ws = {k: torch.as_tensor(v) for k, v in oe.make_synthetic_weights(ecfg).items()}      # checkpoint shapes: the layout heuristic is made for them
    raw = {}
    for k, v in Ws.items():
        raw["model.speaker_encoder." + k] = v.permute(0, 2, 1).contiguous() if v.ndim == 3 else v
    raw["talker.model.norm.weight"] = torch.zeros(4)
    sp = sanitize_speaker_encoder(raw)
    assert set(sp) == {"speaker_encoder." + k for k in Ws}
    for k, v in Ws.items():
        assert tuple(sp["speaker_encoder." + k].shape) == tuple(v.shape) and torch.equal(sp["speaker_encoder." + k], v), k
    assert all(torch.equal(v, sanitize_speaker_encoder({"speaker_encoder." + k: v})["speaker_encoder." + k]) for k, v in Ws.items())   # MLX layout stays
    # configuration: speaker encoder only for base models, tokenizer encoder only with encoder_config
    c = Qwen3TTSConfiguration.from_dict({"tts_model_type": "base", "speaker_encoder_config": {"enc_dim": 2048}},
                                        {"encoder_config": {"num_filters": 32, "upsampling_ratios": [8, 6, 5, 4]}, "encoder_valid_num_quantizers": 16})
    assert c.speaker_encoder.enc_dim == 2048 and c.speaker_encoder.enc_channels == ecfg.enc_channels
    assert c.tokenizer_encoder.num_filters == 32 and c.tokenizer_encoder.upsampling_ratios == (8, 6, 5, 4) and c.encoder_valid_num_quantizers == 16
    c2 = Qwen3TTSConfiguration.from_dict({"tts_model_type": "custom_voice"}, {})
    assert c2.speaker_encoder is None and c2.tokenizer_encoder is None
    rc = c.reference_to_c()
    assert (rc.spk_n_blocks, rc.spk_enc_dim, rc.enc_n_ratios, rc.enc_num_filters, rc.enc_valid_num_quantizers) == (5, 2048, 4, 32, 16)


def test_in_context_prompt_layout_without_a_device():
    """prepareICLGenerationInputs (Qwen3TTS.swift:753-837) as (text id, codec id) positions, with a language id and without a speaker row"""
    cfg = q3.Qwen3TTSConfiguration(codec_language_id={"english": 2050})
    m = object.__new__(q3.Qwen3TTSModel)
    m.configuration = cfg; m.tokenizer = _Tok(); m._h = None
    ctx = q3.ReferenceAudioContext(None, np.zeros((16, 3), np.int32), -1, 7)
    rid = _Tok().encode("<|im_start|>assistant\nabc<|im_end|>\n")[3:-2]
    p = m.prepare_icl_generation_inputs("Hi you", conditioning=(ctx, rid, 2050))
    ids = _Tok().encode("<|im_start|>assistant\nHi you<|im_end|>\n<|im_start|>assistant\n")
    V = cfg.talker.vocab_size
    body = rid + ids[3:-5] + [cfg.tts_eos_token_id]
    assert p.text_ids.tolist() == ids[:3] + [cfg.tts_pad_token_id] * 4 + [cfg.tts_bos_token_id] + body + [cfg.tts_pad_token_id] * 4
    assert p.codec_ids.tolist() == [-1] * 3 + [cfg.codec_think_id, cfg.codec_think_bos_id, 2050, cfg.codec_think_eos_id, cfg.codec_pad_id] \
        + [cfg.codec_pad_id] * len(body) + [cfg.codec_bos_id, V + 7, V + 8, V + 9]
    assert len(p.trailing_ids) == 0 and p.target_token_count == 6 and p.reference is ctx


def test_product_side_front_end_checkpoint_generator_has_the_oracles_key_set():
    """mlx_audio_swift_amd.synthetic.qwen3tts_reference_synthetic_weights (used by tools/bench_q3_reference.py, which may not import the
    oracle) yields exactly the tensors the engine asks for: the oracle generators' keys behind the two prefixes, kept quantizers only."""
    from mlx_audio_swift_amd.synthetic import qwen3tts_reference_synthetic_weights
    from oracle import ecapa as oe
    from oracle import mimi_encoder as om
    cfg = q3.Qwen3TTSConfiguration()
    ec = oe.EcapaConfig(mel_dim=12, enc_dim=20, enc_channels=(16, 16, 16, 16, 48), enc_kernel_sizes=(5, 3, 3, 3, 1), enc_dilations=(1, 2, 3, 4, 1),
                        enc_attention_channels=8, enc_res2net_scale=4, enc_se_channels=6)           # three SE-Res2Net blocks of 16 -> mfa over 48
    cfg.speaker_encoder = q3.Qwen3TTSSpeakerEncoderConfiguration(**{k: getattr(ec, k) for k in q3.Qwen3TTSSpeakerEncoderConfiguration.__dataclass_fields__})
    cfg.tokenizer_encoder = q3.Qwen3TTSTokenizerEncoderConfiguration(num_filters=4, hidden_size=16, num_hidden_layers=2, num_attention_heads=2, intermediate_size=32,
                                                                    codebook_dim=8, codebook_size=32, num_quantizers=5)
    cfg.encoder_valid_num_quantizers = 4
    got = {n: a.shape for n, a in qwen3tts_reference_synthetic_weights(cfg)}
    sp = {"speaker_encoder." + k: np.asarray(v).shape for k, v in oe.make_synthetic_weights(ec).items()}
    mc = om.MimiEncoderConfig(num_filters=4, hidden_size=16, num_hidden_layers=2, num_attention_heads=2, intermediate_size=32, codebook_dim=8,
                              codebook_size=32, num_quantizers=5, valid_num_quantizers=4)
    en = {"encoder_model." + k: np.asarray(v).shape for k, v in om.make_synthetic_weights(mc).items() if "rvq_rest.vq.layers.3." not in k}
    assert got == {**sp, **en}
