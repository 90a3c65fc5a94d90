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
    assert set(back) == set(Wd)
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
