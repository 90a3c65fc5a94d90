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
