"""-m gpu: the A/B switches of the decode step select code that is NOT the default - each of them is held to the default's results here,
so that a measured-and-parked variant cannot rot unnoticed (round 4: `MIS_GRAPH_STEPS`, `MIS_ATTN_PAIR`, `MIS_GLUE_CPT`, `MIS_ARR_*`).

* several decode steps per graph launch: the same kernels in the same order -> the same tokens;
* two K/V tiles requested up front in `k_attn_decode2`: the same arithmetic on the same fragments -> bit-identical logits;
* the glue with one column group per thread against three (d = 3072): per-column arithmetic identical, the row's sum of squares is reduced in
  another order -> logits within the tolerance every LM test uses against the oracle (and far inside it);
* GEMM arrangements per role (`MIS_ARR_TUNED=0`: the round-3 set): other float32 summation orders -> the same tolerance."""
import numpy as np
import pytest

import mlx_audio_swift_amd as mas
from gpu_util import lm_host_config, logits_errors
from oracle import llama as ollama

pytestmark = pytest.mark.gpu


def _teacher_forced_logits(dev, rows, steps_with_logits, max_context):
    B = len(rows)
    dev.lm_reset(B, max_context)
    got = {}
    for t in range(max(len(r) for r in rows)):
        ids = np.asarray([r[t] if t < len(r) else 0 for r in rows], np.int32)
        act = np.asarray([1 if t < len(r) else 0 for r in rows], np.uint8)
        if t in steps_with_logits:
            got[t] = dev.lm_forward(ids, act).copy()
        else:
            dev.lm_forward(ids, act, want_logits=False)
    return got


def test_attention_pair_and_glue_and_arrangement_switches_at_orpheus_width(monkeypatch):
    cfg = ollama.LlamaConfig(num_hidden_layers=2)                       # Orpheus-3B widths: d 3072, 24 / 8 heads x 128, ffn 8192, V 156 940
    dev = mas.LlamaTTSModel.synthetic(lm_host_config(cfg), seed=4321)
    B = 4
    rng = np.random.default_rng(5)
    lens = [330, 300, 40, 290]                                          # > 256 keys: waves with two tiles (the pair variant's case)
    rows = [np.concatenate([[128259], rng.integers(0, 128000, n - 1)]).astype(np.int32) for n in lens]
    steps = {39, 289, 299, 329}
    for k in ("MIS_ATTN_PAIR", "MIS_GLUE_CPT", "MIS_ARR_TUNED"):
        monkeypatch.delenv(k, raising=False)
    base = _teacher_forced_logits(dev, rows, steps, 384)
    monkeypatch.setenv("MIS_ATTN_PAIR", "1")
    pair = _teacher_forced_logits(dev, rows, steps, 384)
    monkeypatch.delenv("MIS_ATTN_PAIR")
    for t in steps:
        assert np.array_equal(base[t], pair[t]), t                      # same fragments, same order: bit-identical
    monkeypatch.setenv("MIS_GLUE_CPT", "1")
    glue1 = _teacher_forced_logits(dev, rows, steps, 384)
    monkeypatch.delenv("MIS_GLUE_CPT")
    monkeypatch.setenv("MIS_ARR_TUNED", "0")
    arr3 = _teacher_forced_logits(dev, rows, steps, 384)                # (lm_reset re-reads the arrangement switches)
    monkeypatch.delenv("MIS_ARR_TUNED")
    for name, other in (("glue", glue1), ("arrangement", arr3)):
        for t in steps:
            live = [b for b in range(B) if t < lens[b]]
            e_max, e_rms, n_sure, agree = logits_errors(other[t][live], base[t][live])
            assert e_max <= 0.016 and e_rms <= 0.008 and agree, (name, t, e_max, e_rms)


def test_several_decode_steps_per_graph_launch_return_the_same_tokens(monkeypatch):
    from mlx_audio_swift_amd.synthetic import snac_synthetic_weights
    snac_small = dict(encoder_dim=4, encoder_rates=[2, 4, 8, 8], decoder_dim=64, decoder_rates=[8, 8, 4, 2], codebook_size=4096, codebook_dim=8,
                      vq_strides=[4, 2, 1])
    snac_cfg = mas.SNACConfig(**snac_small)
    codec = mas.SNAC.from_weights(snac_cfg, snac_synthetic_weights(snac_cfg, seed=1234))
    cfg = mas.LlamaTTSConfiguration(hidden_size=256, num_hidden_layers=2, intermediate_size=512, num_attention_heads=2, num_key_value_heads=1,
                                    head_dim=128, vocab_size=156940, rope_theta=500000.0)
    lm = mas.LlamaTTSModel.synthetic(cfg, codec=codec, seed=77)
    rng = np.random.default_rng(3)
    prompts = [np.asarray([128259] + list(rng.integers(0, 128000, n - 4)) + [128009, 128260, 128257], np.int32) for n in (9, 14, 6)]
    toks = {}
    for steps_per_graph, budget in (("1", 42), ("4", 49), ("8", 56)):    # (different budgets: the step graphs are captured again per setting)
        monkeypatch.setenv("MIS_GRAPH_STEPS", steps_per_graph)
        gp = mas.GenerateParameters(max_tokens=budget, temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=11, frame_constrained=2)
        _, toks[steps_per_graph] = lm.generate_batch(prompts, gp, return_tokens=True)
    monkeypatch.delenv("MIS_GRAPH_STEPS")
    for r in range(len(prompts)):
        assert np.array_equal(toks["1"][r], toks["4"][r][:42]) and np.array_equal(toks["1"][r], toks["8"][r][:42]), r
