"""-m gpu: checkpoint directories through mis_tts_load (LlamaTTSModel.fromModelDirectory, LlamaTTS.swift:942-977): plain bf16
safetensors and MLX affine-quantised ones (uint32 weight + scales + biases, (f)3)."""
import json
import os

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

import mlx_audio_swift_amd as mas
from gpu_util import lm_host_config, teacher_forced
from oracle import llama as ollama
from oracle import mlxquant as mq

pytestmark = pytest.mark.gpu

CFG = ollama.LlamaConfig(hidden_size=256, num_hidden_layers=2, intermediate_size=512, num_attention_heads=2, num_key_value_heads=1,
                         head_dim=128, vocab_size=640, tie_word_embeddings=False)


def _check(pairs):
    for dev_l, ref_l in pairs:
        assert np.abs(dev_l - ref_l).max() <= 0.04 * np.abs(ref_l).max()


def test_plain_and_quantised_directories(tmp_path):
    W = ollama.make_synthetic_weights(CFG, seed=77)
    rng = np.random.default_rng(0)
    rows = [rng.integers(0, CFG.vocab_size, n).astype(np.int32) for n in (11, 4)]
    # ---- plain bf16 directory
    d0 = tmp_path / "plain"; d0.mkdir()
    (d0 / "config.json").write_text(json.dumps(CFG.to_json_dict()))
    save_file({k: v.contiguous() for k, v in W.items()}, str(d0 / "model.safetensors"))
    dev = mas.LlamaTTSModel.from_model_directory(str(d0))
    _check(teacher_forced(ollama.LlamaOracle(CFG, W, round="bf16"), dev, rows))
    # ---- quantised: every Linear + the embedding, 4 bit / group 64, one layer overridden to 8 bit
    for bits in (4, 8):
        dq = tmp_path / f"q{bits}"; dq.mkdir()
        cfgj = CFG.to_json_dict()
        cfgj["quantization"] = {"group_size": 64, "bits": bits, "model.layers.1.mlp.down_proj": {"group_size": 32, "bits": 8}}
        (dq / "config.json").write_text(json.dumps(cfgj))
        tensors, Wd = {}, {}
        for k, v in W.items():
            if v.ndim == 2:
                base = k[: -len(".weight")]
                g, b = (32, 8) if base == "model.layers.1.mlp.down_proj" else (64, bits)
                wq, s, bia = mq.quantize(v.float().numpy(), g, b)
                s16, b16 = torch.from_numpy(s).bfloat16(), torch.from_numpy(bia).bfloat16()
                tensors[k] = torch.from_numpy(wq.view(np.int32)).view(torch.int32)
                tensors[base + ".scales"], tensors[base + ".biases"] = s16, b16
                Wd[k] = torch.from_numpy(mq.dequantize(wq, s16.float().numpy(), b16.float().numpy(), g, b)).bfloat16()
            else:
                tensors[k] = v.contiguous(); Wd[k] = v
        # safetensors has no torch uint32: write the words as I32 and patch the header dtype to U32 like MLX writes it
        path = str(dq / "model.safetensors")
        save_file(tensors, path)
        raw = open(path, "rb").read()
        n = int.from_bytes(raw[:8], "little")
        hdr = json.loads(raw[8:8 + n])
        for k in hdr:
            if k != "__metadata__" and hdr[k]["dtype"] == "I32":
                hdr[k]["dtype"] = "U32"
        hb = json.dumps(hdr, separators=(",", ":")).encode()
        hb += b" " * ((8 - len(hb) % 8) % 8)
        open(path, "wb").write(len(hb).to_bytes(8, "little") + hb + raw[8 + n:])
        devq = mas.LlamaTTSModel.from_model_directory(str(dq))
        # oracle on the dequantised (bf16-rounded) weights: the engine dequantises once at load
        _check(teacher_forced(ollama.LlamaOracle(CFG, Wd, round="bf16"), devq, rows))
        # and through the tensor-level entry point
        m = mas.LlamaTTSModel(lm_host_config(CFG))
        for k, v in W.items():
            if v.ndim == 2:
                base = k[: -len(".weight")]
                g, b = (32, 8) if base == "model.layers.1.mlp.down_proj" else (64, bits)
                m.set_quantized_tensor(k, tensors[k].numpy().view(np.uint32), tensors[base + ".scales"], tensors[base + ".biases"], g, b)
            else:
                m.set_tensor(k, v)
        m.finalize()
        _check(teacher_forced(ollama.LlamaOracle(CFG, Wd, round="bf16"), m, rows))
    with pytest.raises(mas.AudioGenerationError):
        bad = tmp_path / "bad"; bad.mkdir()
        (bad / "config.json").write_text(json.dumps(CFG.to_json_dict()))
        save_file({"model.norm.weight": W["model.norm.weight"].contiguous()}, str(bad / "model.safetensors"))
        mas.LlamaTTSModel.from_model_directory(str(bad))                 # update(verify: .all): missing parameters


QCFG = ollama.LlamaConfig(hidden_size=1024, num_hidden_layers=2, intermediate_size=3072, num_attention_heads=16, num_key_value_heads=8,
                          head_dim=128, vocab_size=3072, tie_word_embeddings=False, rope_scaling=None, rope_plain=True, qk_norm=True,
                          rms_norm_eps=1e-6)


@pytest.mark.parametrize("bits,batch", [(8, 32), (4, 32), (8, 40), (4, 5)], ids=["8bit-b32", "4bit-b32", "8bit-b40", "4bit-b5"])
def test_native_quantised_streaming_matches_the_float32_dequant_oracle(bits, batch):
    """QuantizedLinear / quantizedMatmul (LlamaTTS.swift:958-968, Qwen3TTS.swift:1157-1170): uniformly quantised checkpoints (group
    64, bf16 scales) are streamed as codes and dequantised in registers (csrc/lm_qgemm.hip).  Oracle = the LM on the float32
    weights s*q+b, NOT rounded to bf16 - what quantized_matmul computes (tests/test_oracle_mlxquant.py) - at Qwen3-TTS-0.6B talker
    widths (1024 / 3072, 16/8 heads): K = 1024 -> 16 scale groups (split-K S, 4 waves per item), K = 3072 -> 48; batch 32 / 40 / 5
    -> MT = 2 / 3 / 1 (both register-buffer depths).  Tolerance as test_gpu_lm.py: logits max <= 0.016 max|ref|, rms <= 0.008 (observed 0.0077 / 0.0060)."""
    from gpu_util import logits_errors, record
    W = ollama.make_synthetic_weights(QCFG, seed=99)
    m = mas.LlamaTTSModel(lm_host_config(QCFG))
    W32, W16 = {}, {}
    for k, v in W.items():
        if v.ndim == 2:
            wq, s, bia = mq.quantize(v.float().numpy(), 64, bits)
            s16, b16 = torch.from_numpy(s).bfloat16(), torch.from_numpy(bia).bfloat16()
            m.set_quantized_tensor(k, wq, s16, b16, 64, bits)
            d32 = torch.from_numpy(mq.dequantize(wq, s16.float().numpy(), b16.float().numpy(), 64, bits))
            W16[k] = d32.bfloat16()
            W32[k] = W16[k] if k == "model.embed_tokens.weight" else d32       # QuantizedEmbedding yields model-dtype rows
        else:
            m.set_tensor(k, v); W32[k] = v; W16[k] = v
    m.finalize()
    assert m.native_quant_bits == {"qkv": bits, "o": bits, "gate_up": bits, "down": bits, "lm_head": bits}
    rng = np.random.default_rng(7)
    rows = [rng.integers(0, QCFG.vocab_size, 3 + (b % 4)).astype(np.int32) for b in range(batch)]
    pairs = teacher_forced(ollama.LlamaOracle(QCFG, W32, round="bf16"), m, rows)
    ref16 = teacher_forced(ollama.LlamaOracle(QCFG, W16, round="bf16"), m, rows[:4])
    worst = (0.0, 0.0)
    for dev_l, ref_l in pairs:
        e_max, e_rms, n_sure, agree = logits_errors(dev_l, ref_l)
        worst = (max(worst[0], e_max), max(worst[1], e_rms))
        assert e_max <= 0.016 and e_rms <= 0.008 and agree, (e_max, e_rms)
    # how far the dequantise-at-load arithmetic (bf16-rounded weights) sits from the same device logits, for the record
    r16 = max(logits_errors(d, r)[1] for d, r in ref16)
    record(f"native_quant_{bits}bit_b{batch}", logits_max_rel=worst[0], logits_rms_rel=worst[1], rms_vs_bf16_rounded_weights=r16,
           tol_max=0.016, tol_rms=0.008)


def test_quantised_role_with_a_per_layer_override_falls_back_to_the_dense_copy():
    """One matrix of a role quantised differently (config.json per-layer override) -> that role is dequantised at load, the others
    stream natively; results stay within tolerance of the oracle either way."""
    W = ollama.make_synthetic_weights(CFG, seed=78)
    m = mas.LlamaTTSModel(lm_host_config(CFG))
    Wd = {}
    for k, v in W.items():
        if v.ndim == 2:
            g, b = (32, 8) if k == "model.layers.1.mlp.down_proj.weight" else (64, 4)
            wq, s, bia = mq.quantize(v.float().numpy(), g, b)
            s16, b16 = torch.from_numpy(s).bfloat16(), torch.from_numpy(bia).bfloat16()
            m.set_quantized_tensor(k, wq, s16, b16, g, b)
            Wd[k] = torch.from_numpy(mq.dequantize(wq, s16.float().numpy(), b16.float().numpy(), g, b)).bfloat16()
        else:
            m.set_tensor(k, v); Wd[k] = v
    m.finalize()
    assert m.native_quant_bits == {"qkv": 4, "o": 4, "gate_up": 4, "down": 0, "lm_head": 4}
    rng = np.random.default_rng(1)
    rows = [rng.integers(0, CFG.vocab_size, n).astype(np.int32) for n in (9, 5)]
    _check(teacher_forced(ollama.LlamaOracle(CFG, Wd, round="bf16"), m, rows))


@pytest.mark.parametrize("bits", [8, 4])
def test_batched_prefill_on_code_streamed_weights(bits, monkeypatch):
    """`model(inputIds, cache:)` on a quantised checkpoint: the [positions x rows] pass runs k_gemm_skinny_q on 64-row slices (packed
    rows in, float32 sums out, residual add behind it) instead of falling back to one position at a time.  Against the float32-dequant
    oracle (quantizedMatmul's arithmetic) and against the position-by-position path; a decode step continues behind it.  Ragged rows,
    7 rows x up to 30 positions = 480 (position, row) pairs: seven 64-row slices and one of 32."""
    from gpu_util import logits_errors, record
    W = ollama.make_synthetic_weights(QCFG, seed=99)
    m = mas.LlamaTTSModel(lm_host_config(QCFG))
    W32 = {}
    for k, v in W.items():
        if v.ndim == 2:
            wq, s, bia = mq.quantize(v.float().numpy(), 64, bits)
            s16, b16 = torch.from_numpy(s).bfloat16(), torch.from_numpy(bia).bfloat16()
            m.set_quantized_tensor(k, wq, s16, b16, 64, bits)
            d32 = torch.from_numpy(mq.dequantize(wq, s16.float().numpy(), b16.float().numpy(), 64, bits))
            W32[k] = d32.bfloat16() if k == "model.embed_tokens.weight" else d32
        else:
            m.set_tensor(k, v); W32[k] = v
    m.finalize()
    assert m.native_quant_bits["qkv"] == bits
    oracle = ollama.LlamaOracle(QCFG, W32, round="bf16")
    rng = np.random.default_rng(5)
    lens = [30, 1, 17, 29, 2, 8, 23]
    rows = [rng.integers(0, QCFG.vocab_size, n).astype(np.int32) for n in lens]
    nxt = rng.integers(0, QCFG.vocab_size, len(rows)).astype(np.int32)
    monkeypatch.setenv("MIS_PREFILL_SEQ", "0")
    got = m.lm_prefill(rows, max_context=64)
    got2 = m.lm_forward(nxt)
    monkeypatch.setenv("MIS_PREFILL_SEQ", "1")
    seq = m.lm_prefill(rows, max_context=64)
    seq2 = m.lm_forward(nxt)
    oracle.reset(len(rows))
    ref_all = oracle.forward([np.concatenate([r, nxt[i:i + 1]]) for i, r in enumerate(rows)], logit_positions=[[len(r) - 1, len(r)] for r in rows])
    e_all, d_all, m_all = [], [], []
    for b in range(len(rows)):
        ref = ref_all[b].numpy()
        for dv, sq, rf in ((got[b], seq[b], ref[0]), (got2[b], seq2[b], ref[1])):
            e_max, e_rms, _, agree = logits_errors(dv[None], rf[None])
            assert e_max <= 0.016 and agree, (b, e_max)
            e_all.append(e_rms); m_all.append(e_max)
            d_all.append(float(np.sqrt(np.mean((dv - sq) ** 2)) / np.sqrt(np.mean(sq ** 2))))
    record(f"batched_prefill_{bits}bit_codes", logits_max_rel=max(m_all), logits_rms_rel_worst=max(e_all), logits_rms_rel_mean=float(np.mean(e_all)),
           rms_vs_sequential_worst=max(d_all), tol_max=0.016, tol_rms_mean=0.008, tol_rms_worst=0.016)
    assert np.mean(e_all) <= 0.008 and np.max(e_all) <= 0.016, (np.mean(e_all), np.max(e_all))
    assert np.mean(d_all) <= 0.008 and np.max(d_all) <= 0.016, (np.mean(d_all), np.max(d_all))


def test_streaming_arrangement_behind_the_switch():
    """csrc/lm_qgemm.hip launches the one-shot arrangement (k_gemm_skinny_q1) wherever a wave's K share fits a register buffer - at
    these widths: every role.  MIS_QGEMM_V2=0 sends the same launches through the streaming kernel (k_gemm_skinny_q), which the
    Orpheus-sized gate/up and output projections use; the switch is read once per process, so the 8- and 4-bit batch-32 cases of the
    oracle comparison above run again in a child interpreter with it off."""
    import subprocess
    import sys
    env = dict(os.environ, MIS_QGEMM_V2="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-k",
                        "native_quantised and b32"], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
