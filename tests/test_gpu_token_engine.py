"""-m gpu: the batch-1 token engine (csrc/token_engine.hip - one persistent launch per request on the compute units of one XCD; a
laboratory behind include/mi_speech_debug.h) against the oracle and against the product's own decode step, on the same weights.

Soprano-80M's LM widths (hidden 512, ffn 2304, 4 / 1 heads x 128, q/k-norm, plain RoPE; Soprano.swift:38-199) with 2 layers and a cut
vocabulary, so that the oracle (oracle/llama.py, pinned on HF Qwen3ForCausalLM for this variant) finishes in seconds.  Tolerances are
the LM tests' (bf16 storage at every primitive boundary on both sides, other float32 summation orders): logits max <= 0.016 x scale,
rms <= 0.008 x rms, greedy tokens equal wherever the oracle's top-2 margin exceeds the error."""
import numpy as np
import pytest

import mlx_audio_swift_amd as mas
from gpu_util import lm_pair, logits_errors, record
from oracle import llama as ollama

pytestmark = pytest.mark.gpu

CFG = ollama.LlamaConfig(hidden_size=512, num_hidden_layers=2, intermediate_size=2304, num_attention_heads=4, num_key_value_heads=1,
                         head_dim=128, vocab_size=1200, rope_theta=10000.0, rope_scaling=None, tie_word_embeddings=False, qk_norm=True,
                         rope_plain=True, rms_norm_eps=1e-6)


@pytest.mark.parametrize("xcds", [1, 2, 4, 8])
def test_engine_logits_and_greedy_tokens_match_the_oracle_and_the_launch_chain(xcds):
    W, oracle, dev = lm_pair(CFG)
    rng = np.random.default_rng(7)
    prompt = rng.integers(0, CFG.vocab_size, 40).astype(np.int32)          # > 32 positions: the score loop runs two passes
    n_new = 24
    out = dev.debug_token_engine(prompt, n_new, xcds=xcds, want_logits=True, want_hidden=True)
    nxt = out["next_tokens"]
    seq = np.concatenate([prompt, nxt[len(prompt) - 1:len(prompt) - 1 + n_new]]).astype(np.int32)       # every position the engine processed
    assert len(seq) == len(prompt) + n_new
    # the engine fed itself its own arg-max: position t >= n_prompt saw next_tokens[t - 1]
    oracle.reset(1)
    ref = oracle.forward([seq])[0].numpy()
    e_max, e_rms, n_sure, agree = logits_errors(out["logits"], ref)
    assert e_max <= 0.016 and e_rms <= 0.008 and agree and n_sure > 0, (e_max, e_rms, n_sure)
    assert np.array_equal(out["logits"].argmax(1), nxt)                     # (first index on ties, like np.argmax)
    # the hidden tap (model.norm(h), what Soprano's decoder consumes) against the oracle's
    hid_ref = oracle.last_hidden.numpy()
    h_rms = float(np.sqrt(np.mean((out["hidden"] - hid_ref) ** 2)) / np.sqrt(np.mean(hid_ref ** 2)))
    assert h_rms <= 0.01, h_rms
    # the product's launch chain on the same handle, teacher-forced with the same ids: same rounding points, other summation orders
    dev.lm_reset(1, 128)
    chain = np.stack([dev.lm_forward(seq[t:t + 1])[0] for t in range(len(seq))])
    c_max, c_rms, _, c_agree = logits_errors(out["logits"], chain)
    assert c_max <= 0.016 and c_rms <= 0.008 and c_agree, (c_max, c_rms)
    record(f"token_engine_{xcds}xcd", logits_max_rel=e_max, logits_rms_rel=e_rms, vs_launch_chain_max_rel=c_max, vs_launch_chain_rms_rel=c_rms,
           tol_max=0.016, tol_rms=0.008, ms_per_position=out["ms"] / len(seq))
    # deterministic: the same request again gives the same bits
    again = dev.debug_token_engine(prompt, n_new, xcds=xcds, want_logits=True)
    assert np.array_equal(again["logits"], out["logits"]) and np.array_equal(again["next_tokens"], nxt)


def test_engine_at_contexts_beyond_the_prefetched_tiles():
    """The attention phase requests a wave's first two key tiles and the first four 32-key value steps ahead of the layer's first poll
    (128 positions); longer contexts take the loops behind them, and the score rows in LDS hold 1 024 positions.  300 positions here: logits
    at the positions around every boundary (128 keys, 256 keys, the 16- and 32-key tile edges) against the oracle."""
    W, oracle, dev = lm_pair(CFG)
    rng = np.random.default_rng(13)
    prompt = rng.integers(0, CFG.vocab_size, 280).astype(np.int32)
    out = dev.debug_token_engine(prompt, 20, xcds=4, want_logits=True)
    seq = np.concatenate([prompt, out["next_tokens"][len(prompt) - 1:len(prompt) - 1 + 20]]).astype(np.int32)
    keep = sorted({0, 15, 16, 17, 31, 32, 33, 63, 64, 127, 128, 129, 143, 144, 145, 159, 160, 161, 255, 256, 257, 271, 272, 279, 280, 299})
    oracle.reset(1)
    ref = oracle.forward([seq], logit_positions=[keep])[0].numpy()
    e_max, e_rms, n_sure, agree = logits_errors(out["logits"][keep], ref)
    assert e_max <= 0.016 and e_rms <= 0.008 and agree and n_sure > 0, (e_max, e_rms, n_sure)
    record("token_engine_long_context", logits_max_rel=e_max, logits_rms_rel=e_rms, tol_max=0.016, tol_rms=0.008, positions=len(seq))


def test_engine_rejects_other_shapes():
    cfg = ollama.LlamaConfig(**{**ollama.TINY_QWEN3.__dict__})
    W, oracle, dev = lm_pair(cfg)
    with pytest.raises(mas.AudioGenerationError):
        dev.debug_token_engine(np.asarray([1, 2, 3], np.int32), 2)


def test_generate_form_sampler_is_the_oracles_on_the_engines_own_logits():
    """The generate form (what mis_soprano_generate runs at batch 1): a token after the last prompt position and after every generated
    one, chosen INSIDE the persistent launch - temperature 0: arg-max behind the Soprano repetition penalty; else mis-sampler-v1 behind
    it (three more all-to-all edges: maximum, tile masses, token).  Given the logits a token was drawn from (logits row k), the choice
    must be oracle/sampler.py's bit for bit - the same integers as csrc/lm_sampler.hip; the hidden rows are the oracle's; a stop id ends
    the request at once; and the number of XCDs changes nothing (same arithmetic, other placement)."""
    from oracle import sampler as osamp
    from oracle import soprano as osop
    W, oracle, dev = lm_pair(CFG)
    rng = np.random.default_rng(11)
    prompt = rng.integers(0, CFG.vocab_size, 21).astype(np.int32)
    n_new = 34
    for temp in (0.0, 0.7):
        gp = mas.GenerateParameters(max_tokens=n_new, temperature=temp, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30, seed=5,
                                    row_offset=3, sampler_flavor=1)
        out = dev.debug_token_engine(prompt, n_new, xcds=2, want_logits=True, want_hidden=True, sampling=gp)
        assert out["chosen"] == n_new and out["positions"] == len(prompt) + n_new
        toks = out["next_tokens"][len(prompt) - 1:len(prompt) - 1 + n_new]
        for k in range(n_new):
            l = osop.soprano_repetition_penalty(out["logits"][k], list(toks[:k])[-30:], 1.5)
            want = osamp.sample(l, temp, 1.0, 5, 3, k)
            assert toks[k] == want, (temp, k)
        # hidden rows: position n_prompt - 1 + k through the oracle (teacher-forced with the engine's own ids)
        seq = np.concatenate([prompt, toks]).astype(np.int32)
        oracle.reset(1)
        ref_l = oracle.forward([seq])[0].numpy()
        hid_ref = oracle.last_hidden.numpy()[len(prompt) - 1:]
        assert out["hidden"].shape == hid_ref.shape
        assert float(np.sqrt(np.mean((out["hidden"] - hid_ref) ** 2)) / np.sqrt(np.mean(hid_ref ** 2))) <= 0.01
        e_max, e_rms, _, _ = logits_errors(out["logits"], ref_l[len(prompt) - 1:len(prompt) - 1 + n_new])
        assert e_max <= 0.016 and e_rms <= 0.008, (e_max, e_rms)
        for xcds in (1, 4):
            other = dev.debug_token_engine(prompt, n_new, xcds=xcds, want_logits=True, sampling=gp)
            assert np.array_equal(other["next_tokens"], out["next_tokens"]) and np.array_equal(other["logits"], out["logits"]), (temp, xcds)
        # a stop id: the fifth chosen id (if it has not come up before) ends the request there - it is reported, nothing follows it
        stop = int(toks[4])
        if stop not in toks[:4]:
            st = dev.debug_token_engine(prompt, n_new, xcds=2, want_hidden=True, sampling=gp, stop_id=stop)
            assert st["chosen"] == 5 and st["positions"] == len(prompt) + 4
            assert np.array_equal(st["next_tokens"][len(prompt) - 1:len(prompt) + 4], toks[:5])
            assert np.array_equal(st["hidden"][:5], out["hidden"][:5])
