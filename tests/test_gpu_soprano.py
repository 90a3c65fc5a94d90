"""-m gpu: Soprano (a21): Vocos/ISTFT decoder against oracle/soprano.py, the Soprano sampler flavour against the
oracle sampler, and generate() end to end (hidden-state collection, [STOP] handling, ragged rows).
Decoder tolerance: float32 chain with exp/cos/sin and a 2*(n_fft/2+1)-term inverse DFT: max |err| <= 2e-4 * max |ref|."""
import numpy as np
import pytest
import torch

import mlx_audio_swift_amd as mas
from gpu_util import lm_host_config
from oracle import llama as ollama
from oracle import sampler as osamp
from oracle import soprano as osop

pytestmark = pytest.mark.gpu

LM = ollama.TINY_QWEN3


def _cfg(**dec):
    base = dict(decoder_num_layers=2, decoder_dim=96, decoder_intermediate_dim=160, hop_length=32, n_fft=128, upscale=4,
                input_kernel=3, dw_kernel=3, token_size=128)
    base.update(dec)
    h = lm_host_config(LM)
    cfg = mas.SopranoConfiguration(hidden_size=LM.hidden_size, num_hidden_layers=LM.num_hidden_layers,
                                   intermediate_size=LM.intermediate_size, num_attention_heads=LM.num_attention_heads,
                                   num_key_value_heads=LM.num_key_value_heads, head_dim=LM.head_dim, vocab_size=LM.vocab_size,
                                   rms_norm_eps=LM.rms_norm_eps, rope_theta=LM.rope_theta, tie_word_embeddings=False,
                                   stop_token_id=3, **base)
    ocfg = osop.SopranoDecoderConfig(hidden_size=LM.hidden_size, **base)
    return cfg, ocfg


def _pair(stop_token_id=3, **dec):
    cfg, ocfg = _cfg(**dec)
    cfg.stop_token_id = stop_token_id
    Wd = osop.make_synthetic_weights(ocfg, seed=99)
    Wl = ollama.make_synthetic_weights(LM, seed=4321)
    # checkpoint naming: LM keys without the "model." prefix exercise sanitize (Soprano.swift:314-361)
    W = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in Wl.items()}
    W.update(Wd)
    dev = mas.SopranoModel.from_weights(cfg, W)
    return cfg, dev, osop.SopranoDecoderOracle(ocfg, Wd), ollama.LlamaOracle(LM, Wl, round="bf16")


@pytest.mark.parametrize("dec", [dict(), dict(input_kernel=1, dw_kernel=7, n_fft=64, hop_length=16, upscale=2, token_size=32)],
                         ids=["k3-dw3-up4", "k1-dw7-up2"])
def test_decoder_matches_oracle(dec):
    cfg, dev, odec, _ = _pair(**dec)
    rng = np.random.default_rng(0)
    for B, L in [(2, 9), (1, 2), (3, 33)]:
        hid = rng.standard_normal((B, L, cfg.hidden_size)).astype(np.float32)
        ref = odec.decode(hid)
        got = dev.decode(hid)
        assert got.shape == ref.shape == (B, cfg.upscale * (L - 1) * cfg.hop_length)
        assert np.abs(got - ref).max() <= 2e-4 * np.abs(ref).max(), (B, L)
    assert dev.decode(rng.standard_normal((2, 1, cfg.hidden_size)).astype(np.float32)).shape == (2, 0)   # one hidden state: no audio
    # product synthetic generator == oracle generator (shared mis-synth-v1)
    from mlx_audio_swift_amd.synthetic import soprano_decoder_synthetic_weights
    Wp = soprano_decoder_synthetic_weights(cfg, 99)
    Wo = osop.make_synthetic_weights(_cfg(**dec)[1], seed=99)
    assert Wp.keys() == Wo.keys() and all(np.array_equal(Wp[k], Wo[k]) for k in Wp)


def test_soprano_sampler_flavour_matches_oracle():
    rng = np.random.default_rng(1)
    V, B, ctx = 1200, 4, 30
    logits = osamp.bf16_round((rng.standard_normal((B, V)) * 3).astype(np.float32)) if hasattr(osamp, "bf16_round") else None
    if logits is None:
        logits = torch.as_tensor((rng.standard_normal((B, V)) * 3).astype(np.float32)).bfloat16().float().numpy()
    win = np.zeros((B, ctx), np.int32)
    wl = np.asarray([0, 5, 30, 12], np.int32)
    gen = []
    for b in range(B):
        ids = rng.integers(0, V, wl[b])
        if wl[b] >= 5:
            ids[1] = ids[3] = ids[4]                       # repeated ids: the penalty applies once per occurrence
        ids[: min(2, wl[b])] = np.argsort(logits[b])[-2:][: min(2, wl[b])]   # penalise the top logits so it matters
        gen.append(ids)
        win[b, ctx - wl[b]:] = ids
    for temp in (0.0, 0.7):
        gp = mas.GenerateParameters(temperature=temp, top_p=0.95, repetition_penalty=1.5, repetition_context_size=ctx, seed=5,
                                    sampler_flavor=1)
        got = mas.tts.sample_logits(logits, win, wl, gp, step=7)
        for b in range(B):
            l = osop.soprano_repetition_penalty(logits[b], gen[b], 1.5)
            ref = osamp.sample(l, temp, 1.0, 5, b, 7)
            assert got[b] == ref, (temp, b)


def _teacher_hidden(dev, prompt, toks):
    """Hidden states the engine's own LM produces for prompt + tokens (B = 1): the prompt through the prefill entry (what generate
    runs: a row's result does not depend on the rows beside it), then one decode step per token via the lm_forward tap."""
    _, hid0 = dev.lm.lm_prefill([np.asarray(prompt, np.int32)], max_context=128, want_logits=False, want_hidden=True)
    out = [hid0[0].copy()]
    for t in toks:
        _, hid = dev.lm.lm_forward(np.asarray([t], np.int32), want_hidden=True)
        out.append(hid[0].copy())
    return np.stack(out)[None]


def test_generate_greedy_hidden_collection_stop_and_ragged_rows():
    cfg, dev, odec, olm = _pair()
    rng = np.random.default_rng(2)
    prompts = [rng.integers(4, LM.vocab_size, n).astype(np.int32) for n in (7, 12, 5)]
    gp = mas.GenerateParameters(max_tokens=12, temperature=0.0, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30,
                                seed=1, sampler_flavor=1)
    pcm, toks = dev.generate_batch(prompts, gp, return_tokens=True)
    assert [len(t) for t in toks] == [12, 12, 12]
    # (1) engine tokens are the oracle's greedy choice under teacher forcing (Soprano penalty on generated tokens only)
    olm.reset(3)
    for b in range(3):
        seq = list(prompts[b]) + list(toks[b])
        lg = olm._forward_row(b, torch.as_tensor(np.asarray(seq[:-1], np.int64))).numpy()
        tol = 0.04 * float(np.abs(lg).max())
        for i, t in enumerate(toks[b]):
            l = osop.soprano_repetition_penalty(lg[len(prompts[b]) - 1 + i], list(toks[b][:i])[-30:], 1.5)
            assert l[t] >= l.max() - tol, (b, i)
    # (2) audio == oracle decoder applied to the hidden states of (last prompt token, each generated token)
    for b in range(3):
        hid = _teacher_hidden(dev, prompts[b], toks[b])
        assert hid.shape[1] == 13
        ref = odec.decode(hid)[0]
        assert pcm[b].shape == ref.shape == (12 * cfg.token_size,)
        assert np.abs(pcm[b] - ref).max() <= 2e-4 * np.abs(ref).max(), b
    # (3) determinism and batch == single
    pcm2, toks2 = dev.generate_batch(prompts, gp, return_tokens=True)
    assert all(np.array_equal(a, c) for a, c in zip(toks, toks2)) and all(np.array_equal(a, c) for a, c in zip(pcm, pcm2))
    p1, t1 = dev.generate_batch(prompts[1:2], gp, return_tokens=True)
    assert np.array_equal(t1[0], toks[1]) and np.array_equal(p1[0], pcm[1])
    # (4) [STOP]: choose row 0's 5th token as the stop id -> row 0 ends there, the others run on (ragged batch)
    stop = int(toks[0][4])
    assert stop not in toks[0][:4]
    cfg_s, dev_s, _, _ = _pair(stop_token_id=stop)
    pcm_s, toks_s = dev_s.generate_batch(prompts, gp, return_tokens=True)
    assert np.array_equal(toks_s[0], toks[0][:4])                      # the stop token is not emitted (Soprano.swift:855-857)
    ref0 = odec.decode(_teacher_hidden(dev_s, prompts[0], toks_s[0]))[0]
    assert np.abs(pcm_s[0] - ref0).max() <= 2e-4 * np.abs(ref0).max()
    assert len(pcm_s[0]) == 4 * cfg.token_size
    for b in (1, 2):
        k = list(toks[b]).index(stop) if stop in toks[b] else 12
        assert np.array_equal(toks_s[b], toks[b][:k]) and len(pcm_s[b]) == k * cfg.token_size


def test_generate_sampled_is_seeded_and_text_front_end():
    cfg, dev, _, _ = _pair()

    class Tok:                                                         # stand-in tokenizer: one id per character
        def encode(self, s):
            return {"[STOP]": [3], "[TEXT]": [1], "[START]": [2]}.get(s, [10 + (ord(ch) % 500) for ch in s])

    dev.tokenizer = Tok()
    cfg.space_token_id = 9
    ids = dev.tokenize("[STOP][TEXT]ab  c.[START]")
    assert list(ids) == [3, 1, 10 + ord("a"), 10 + ord("b"), 9, 9, 10 + ord("c"), 10 + ord("."), 2]
    gp = mas.GenerateParameters(max_tokens=6, temperature=0.7, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30,
                                seed=21, sampler_flavor=1)
    a = dev.generate("Hello there. This sentence is long enough to stand alone, yes.\nSecond line here.", generation_parameters=gp)
    b = dev.generate("Hello there. This sentence is long enough to stand alone, yes.\nSecond line here.", generation_parameters=gp)
    assert a.shape == (2 * 6 * cfg.token_size,) and np.array_equal(a, b)    # 2 sentence prompts (short one merged), 6 tokens each
    gp.seed = 22
    c = dev.generate("Hello there. This sentence is long enough to stand alone, yes.\nSecond line here.", generation_parameters=gp)
    assert not np.array_equal(a, c)
    with pytest.raises(mas.AudioGenerationError):
        dev.generate("   ", generation_parameters=gp)


def test_generate_stream_event_order_and_cancel():
    # generateStream (Soprano.swift:693-800): .token* while generating (no [STOP]), then per row .info and ONE .audio whose
    # samples equal generate()'s; closing the stream cancels the engine
    cfg, dev, odec, olm = _pair()
    rng = np.random.default_rng(5)
    prompts = [rng.integers(4, LM.vocab_size, n).astype(np.int32) for n in (6, 9)]
    gp = mas.GenerateParameters(max_tokens=20, temperature=0.0, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30,
                                seed=3, sampler_flavor=1)
    pcm, toks = dev.generate_batch(prompts, gp, return_tokens=True)
    ev = list(dev.generate_stream_batch(prompts, gp))
    for row in (0, 1):
        kinds = [type(e).__name__ for e in ev if e.row == row]
        assert kinds == ["TokenEvent"] * len(toks[row]) + ["InfoEvent", "AudioEvent"]
        assert np.array_equal([e.token for e in ev if e.row == row and isinstance(e, mas.TokenEvent)], toks[row])
        audio = [e.audio for e in ev if e.row == row and isinstance(e, mas.AudioEvent)][0]
        assert np.array_equal(audio, pcm[row])
        info = [e.info for e in ev if e.row == row and isinstance(e, mas.InfoEvent)][0]
        assert info.prompt_token_count == 0 and info.generation_token_count == len(toks[row]) + 1      # hidden states (:752)
    # every token precedes every info/audio event (tokens are delivered while the loop runs)
    last_tok = max(i for i, e in enumerate(ev) if isinstance(e, mas.TokenEvent))
    first_end = min(i for i, e in enumerate(ev) if not isinstance(e, mas.TokenEvent))
    assert last_tok < first_end
    # early close: the engine is cancelled, no error surfaces, and the model stays usable
    long_gp = mas.GenerateParameters(max_tokens=400, temperature=0.0, repetition_penalty=1.5, repetition_context_size=30, sampler_flavor=1)
    g = dev.generate_stream_batch(prompts, long_gp)
    first = next(g)
    assert isinstance(first, mas.TokenEvent)
    g.close()
    pcm2, toks2 = dev.generate_batch(prompts, gp, return_tokens=True)
    assert all(np.array_equal(a, b) for a, b in zip(toks, toks2))
    import ctypes as C
    with pytest.raises(mas.AudioGenerationError) as e:
        list(dev.generate_stream_batch(prompts, long_gp, cancel_flag=C.c_int(1)))
    assert e.value.case == "cancelled"


def test_group_of_two_logical_shards_returns_the_unsharded_audio():
    """mis_soprano_group_generate: sentence prompts sharded over two replicas (same weights, one GPU) - tokens and samples of every
    row equal the single-handle call (sampler RNG keyed by the global row)."""
    cfg, dev, _, _ = _pair()
    _, dev2, _, _ = _pair()
    rng = np.random.default_rng(2)
    rows = [rng.integers(10, 400, n).astype(np.int32) for n in (7, 12, 5, 9, 6)]
    gp = mas.GenerateParameters(max_tokens=6, temperature=0.7, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30, seed=3,
                                sampler_flavor=1)
    pcm, toks = dev.generate_batch(rows, gp, return_tokens=True)
    pcm2, toks2 = dev.generate_batch(rows, gp, return_tokens=True, replicas=[dev, dev2])
    for r in range(len(rows)):
        assert np.array_equal(toks[r], toks2[r]) and np.array_equal(pcm[r], pcm2[r]), r


ENGINE_LM = ollama.LlamaConfig(hidden_size=512, num_hidden_layers=2, intermediate_size=2304, num_attention_heads=4, num_key_value_heads=1,
                               head_dim=128, vocab_size=1200, rope_theta=10000.0, rope_scaling=None, tie_word_embeddings=False, qk_norm=True,
                               rope_plain=True, rms_norm_eps=1e-6)
ENGINE_DEC = dict(decoder_num_layers=2, decoder_dim=96, decoder_intermediate_dim=160, hop_length=32, n_fft=128, upscale=4, input_kernel=3,
                  dw_kernel=3, token_size=128)


def _engine_pair(stop):
    """A Soprano model whose LM has the widths the batch-1 token engine is compiled for (Soprano-80M's; two layers, cut vocabulary)."""
    lm, base = ENGINE_LM, ENGINE_DEC
    cfg = mas.SopranoConfiguration(hidden_size=lm.hidden_size, num_hidden_layers=lm.num_hidden_layers, intermediate_size=lm.intermediate_size,
                                   num_attention_heads=lm.num_attention_heads, num_key_value_heads=lm.num_key_value_heads, head_dim=lm.head_dim,
                                   vocab_size=lm.vocab_size, rms_norm_eps=lm.rms_norm_eps, rope_theta=lm.rope_theta, tie_word_embeddings=False,
                                   stop_token_id=stop, **base)
    ocfg = osop.SopranoDecoderConfig(hidden_size=lm.hidden_size, **base)
    Wd = osop.make_synthetic_weights(ocfg, seed=99)
    Wl = ollama.make_synthetic_weights(lm, seed=4321)
    Wall = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in Wl.items()}
    Wall.update(Wd)
    return cfg, mas.SopranoModel.from_weights(cfg, Wall), osop.SopranoDecoderOracle(ocfg, Wd), ollama.LlamaOracle(lm, Wl, round="bf16")


def test_batch_one_generate_runs_on_the_token_engine(monkeypatch):
    """mis_soprano_generate at batch 1 on an LM of Soprano-80M's widths (hidden 512, ffn 2304, 4 / 1 heads x 128; two layers here): the LM
    loop is ONE persistent launch (csrc/token_engine.hip) instead of the launch chain.  Held to the same bar as the chain
    (test_generate_greedy...): tokens = the oracle's choice under teacher forcing within the logit tolerance, audio = the oracle decoder on
    the hidden states of (last prompt token, every generated token), [STOP] ends the row unannounced; sampling is seeded; and the chain
    (MIS_TOKEN_ENGINE=0) on the same handle gives audio of the same length built from ITS hidden states (the two LMs agree to the logit
    tolerance, not bit for bit: other float32 summation orders)."""
    lm, build = ENGINE_LM, _engine_pair
    cfg, dev, odec, olm = build(3)
    lib = mas._lib.lib()
    rng = np.random.default_rng(2)
    prompt = rng.integers(4, lm.vocab_size, 9).astype(np.int32)
    gp = mas.GenerateParameters(max_tokens=12, temperature=0.0, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30, seed=1,
                                sampler_flavor=1)
    monkeypatch.delenv("MIS_TOKEN_ENGINE", raising=False)
    pcm, toks = dev.generate_batch([prompt], gp, return_tokens=True)
    assert dev.lm_path == 1                                                   # (mis_soprano_lm_path: the engine ran the LM loop)
    assert len(toks[0]) == 12 and pcm[0].shape == (12 * cfg.token_size,)
    # (1) tokens: the oracle's greedy choice under teacher forcing (penalty over the generated ids only), within the logit tolerance
    olm.reset(1)
    seq = list(prompt) + list(toks[0])
    lg = olm._forward_row(0, torch.as_tensor(np.asarray(seq[:-1], np.int64))).numpy()
    tol = 0.04 * float(np.abs(lg).max())
    for i, t in enumerate(toks[0]):
        l = osop.soprano_repetition_penalty(lg[len(prompt) - 1 + i], list(toks[0][:i])[-30:], 1.5)
        assert l[t] >= l.max() - tol, i
    # (2) audio = the oracle decoder on the engine's own hidden rows (the same request through the debug entry point gives them), and those
    # rows = the oracle LM's under teacher forcing within 1 % rms (tests/test_gpu_token_engine.py holds them to that at more positions)
    eng = dev.lm.debug_token_engine(prompt, 12, xcds=2, want_hidden=True, sampling=gp, stop_id=3)
    assert np.array_equal(eng["next_tokens"][len(prompt) - 1:len(prompt) + 11], toks[0]) and eng["hidden"].shape == (13, lm.hidden_size)
    ref = odec.decode(eng["hidden"][None])[0]
    assert pcm[0].shape == ref.shape and np.abs(pcm[0] - ref).max() <= 2e-4 * np.abs(ref).max()
    olm.reset(1)
    olm._forward_row(0, torch.as_tensor(np.asarray(seq, np.int64)))
    hid_ref = olm.last_hidden.numpy()[len(prompt) - 1:]
    assert float(np.sqrt(np.mean((eng["hidden"] - hid_ref) ** 2)) / np.sqrt(np.mean(hid_ref ** 2))) <= 0.01
    # (3) deterministic; the launch chain on the same handle: same length, audio close (its own hidden states)
    pcm2, toks2 = dev.generate_batch([prompt], gp, return_tokens=True)
    assert np.array_equal(toks2[0], toks[0]) and np.array_equal(pcm2[0], pcm[0])
    monkeypatch.setenv("MIS_TOKEN_ENGINE", "0")
    pcm_c, toks_c = dev.generate_batch([prompt], gp, return_tokens=True)
    assert dev.lm_path == 0
    monkeypatch.delenv("MIS_TOKEN_ENGINE")
    assert len(toks_c[0]) == 12 and pcm_c[0].shape == pcm[0].shape
    # (token-for-token equality of the two LM loops is not asserted: they agree to the logit tolerance, and a near-tie may send them apart)
    # (4) [STOP]: the fifth token as the stop id ends the row there, unannounced
    stop = int(toks[0][4])
    if stop not in toks[0][:4]:
        cfg_s, dev_s, _, _ = build(stop)
        pcm_s, toks_s = dev_s.generate_batch([prompt], gp, return_tokens=True)
        assert np.array_equal(toks_s[0], toks[0][:4]) and len(pcm_s[0]) == 4 * cfg.token_size
        ref_s = odec.decode(eng["hidden"][None, :5])[0]                        # five hidden rows: the last prompt token + four generated ones
        want = ref_s[len(ref_s) - 4 * cfg.token_size:]                        # audio[-(n - 1) * token_size:], Soprano.swift:666-671
        assert np.abs(pcm_s[0] - want).max() <= 2e-4 * np.abs(want).max()
    # (5) sampling: seeded
    gs = mas.GenerateParameters(max_tokens=10, temperature=0.7, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30, seed=21,
                                sampler_flavor=1)
    a, ta = dev.generate_batch([prompt], gs, return_tokens=True)
    b, tb = dev.generate_batch([prompt], gs, return_tokens=True)
    assert np.array_equal(ta[0], tb[0]) and np.array_equal(a[0], b[0])
    gs.seed = 22
    c, tc = dev.generate_batch([prompt], gs, return_tokens=True)
    assert not np.array_equal(ta[0], tc[0])


def test_batch_one_stream_runs_on_the_token_engine_and_equals_generate(monkeypatch):
    """generateStream at ONE row (the entry point the reference's CLI times, App.swift:130-138; its generate is built on streamGenerate,
    Soprano.swift:801-885) runs on the SAME persistent launch as generate: the ids reach host-visible memory one store each while the
    launch runs and the calling thread fires .token from there.  Stream and non-stream: same ids, same samples (greedy and sampled);
    the events arrive WHILE the launch runs; a cancel flag ends the launch at its next position."""
    import ctypes as C
    import time
    monkeypatch.delenv("MIS_TOKEN_ENGINE", raising=False)
    cfg, dev, _, _ = _engine_pair(3)
    prompt = np.random.default_rng(2).integers(4, ENGINE_LM.vocab_size, 9).astype(np.int32)
    for temp in (0.0, 0.7):
        gp = mas.GenerateParameters(max_tokens=40, temperature=temp, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30, seed=5,
                                    sampler_flavor=1)
        pcm, toks = dev.generate_batch([prompt], gp, return_tokens=True)
        assert dev.lm_path == 1
        ev = list(dev.generate_stream_batch([prompt], gp))
        assert dev.lm_path == 1
        assert [type(e).__name__ for e in ev] == ["TokenEvent"] * len(toks[0]) + ["InfoEvent", "AudioEvent"]
        assert np.array_equal([e.token for e in ev if isinstance(e, mas.TokenEvent)], toks[0]), temp
        assert np.array_equal(ev[-1].audio, pcm[0]), temp
        assert ev[-2].info.generation_token_count == len(toks[0]) + 1 and ev[-2].info.generate_time > 0
    # the ids are delivered while the launch runs: over a 900-position request the first and the last .token are most of the call apart
    long_gp = mas.GenerateParameters(max_tokens=900, temperature=0.0, repetition_penalty=1.5, repetition_context_size=30, sampler_flavor=1)
    stamps = []
    t0 = time.perf_counter()
    for e in dev.generate_stream_batch([prompt], long_gp):
        stamps.append((type(e).__name__, time.perf_counter() - t0))
    tok_t = [t for k, t in stamps if k == "TokenEvent"]
    assert len(tok_t) >= 100
    from gpu_util import record
    record("soprano_stream_on_token_engine", tokens=len(tok_t), first_token_s=tok_t[0], last_token_s=tok_t[-1], call_s=stamps[-1][1])
    assert tok_t[-1] - tok_t[0] >= 0.3 * tok_t[-1], (tok_t[0], tok_t[-1])
    # early close -> the cancel word reaches the launch, which ends at its next position; no error surfaces, the handle stays usable
    g = dev.generate_stream_batch([prompt], long_gp)
    assert isinstance(next(g), mas.TokenEvent)
    t1 = time.perf_counter()
    g.close()
    closed_in = time.perf_counter() - t1
    gp = mas.GenerateParameters(max_tokens=40, temperature=0.0, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30, seed=5, sampler_flavor=1)
    pcm2, toks2 = dev.generate_batch([prompt], gp, return_tokens=True)
    ev2 = list(dev.generate_stream_batch([prompt], gp))
    assert np.array_equal([e.token for e in ev2 if isinstance(e, mas.TokenEvent)], toks2[0]) and np.array_equal(ev2[-1].audio, pcm2[0])
    with pytest.raises(mas.AudioGenerationError) as err:
        list(dev.generate_stream_batch([prompt], long_gp, cancel_flag=C.c_int(1)))
    assert err.value.case == "cancelled"
    record("soprano_stream_cancel_on_token_engine", close_returned_in_s=closed_in)


def test_token_engine_time_out_falls_back_to_the_launch_chain_and_reports_it(monkeypatch):
    """the engine -> launch-chain fall-back had no test, and was silent.  MIS_TE_SPIN=0 makes the first granule poll that misses
    count as a time-out (what happens when another stream holds compute units and the workers cannot all be resident): the request then
    runs on the launch chain - ids and samples equal the chain's own (MIS_TOKEN_ENGINE=0), nothing is announced twice in the stream form -
    and mis_soprano_lm_path() says 2, not 1 or 0."""
    cfg, dev, _, _ = _engine_pair(3)
    prompt = np.random.default_rng(3).integers(4, ENGINE_LM.vocab_size, 11).astype(np.int32)
    gp = mas.GenerateParameters(max_tokens=16, temperature=0.7, top_p=0.95, repetition_penalty=1.5, repetition_context_size=30, seed=8,
                                sampler_flavor=1)
    monkeypatch.setenv("MIS_TOKEN_ENGINE", "0")
    pcm_c, toks_c = dev.generate_batch([prompt], gp, return_tokens=True)
    assert dev.lm_path == 0
    monkeypatch.delenv("MIS_TOKEN_ENGINE")
    monkeypatch.setenv("MIS_TE_SPIN", "0")
    pcm_f, toks_f = dev.generate_batch([prompt], gp, return_tokens=True)
    assert dev.lm_path == 2
    assert np.array_equal(toks_f[0], toks_c[0]) and np.array_equal(pcm_f[0], pcm_c[0])
    ev = list(dev.generate_stream_batch([prompt], gp))
    assert dev.lm_path == 2
    assert np.array_equal([e.token for e in ev if isinstance(e, mas.TokenEvent)], toks_c[0]) and np.array_equal(ev[-1].audio, pcm_c[0])
    monkeypatch.delenv("MIS_TE_SPIN")
    pcm_e, toks_e = dev.generate_batch([prompt], gp, return_tokens=True)
    assert dev.lm_path == 1 and len(pcm_e[0]) > 0
