"""CPU: pins for oracle/qwen3tts.py (a22).  The reference cannot run here ("parity unpinned"); these tests pin the
restatement against independent formulations and the structural facts the reference relies on."""
import numpy as np
import torch

from oracle import qwen3tts as oq


def test_causal_transposed_conv_matches_naive_scatter():
    # ConvTransposed1d(padding 0) + right trim (Qwen3TTSSpeechTokenizer.swift:533-551): y[n*s + j] += x[c, n] * w[o, j, c]
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 3, 5)).astype(np.float32)
    for s, k in ((2, 4), (3, 6), (2, 2)):
        w = rng.standard_normal((4, k, 3)).astype(np.float32)
        b = rng.standard_normal(4).astype(np.float32)
        ref = np.zeros((4, (5 - 1) * s + k), np.float64)
        for n in range(5):
            for j in range(k):
                ref[:, n * s + j] += w[:, j, :].astype(np.float64) @ x[0, :, n]
        ref = (ref + b[:, None])[:, : 5 * s]
        got = oq.causal_conv_transpose1d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), s).numpy()[0]
        assert got.shape == (4, 5 * s) and np.abs(got - ref).max() < 1e-5


def test_decoder_is_causal_so_streaming_equals_full_decode():
    cfg = oq.TINY.decoder
    dec = oq.SpeechDecoderOracle(cfg, oq.make_synthetic_decoder_weights(cfg))
    rng = np.random.default_rng(1)
    codes = rng.integers(0, cfg.codebook_size, (2, cfg.num_quantizers, 9))
    full = dec.decode(codes)
    assert full.shape == (2, 9 * cfg.total_upsample) and np.abs(full).max() <= 1.0 and np.abs(full).std() > 1e-3
    for n in (1, 4):                              # a prefix of the frames gives exactly the prefix of the waveform
        part = dec.decode(codes[:, :, :n])
        assert np.abs(part - full[:, : n * cfg.total_upsample]).max() < 2e-5
    changed = codes.copy(); changed[:, :, 6] = (changed[:, :, 6] + 1) % cfg.codebook_size
    alt = dec.decode(changed)
    assert np.array_equal(alt[:, : 6 * cfg.total_upsample], full[:, : 6 * cfg.total_upsample]) or \
        np.abs(alt[:, : 6 * cfg.total_upsample] - full[:, : 6 * cfg.total_upsample]).max() < 2e-5
    assert np.abs(alt - full).max() > 1e-4


def test_sample_token_set_semantics():
    rng = np.random.default_rng(2)
    V = 300
    lg = torch.as_tensor((rng.standard_normal(V) * 3).astype(np.float32)).bfloat16().float().numpy()
    top = np.argsort(lg)[::-1]
    sup = list(range(V - 50, V))
    # greedy honours suppress + penalty
    lg2 = lg.copy(); lg2[V - 1] = 50.0
    assert oq.sample_token(lg2, 0.0, 1.0, 0, 1.0, None, sup, None, 0.0, 1, 0, 0) == int(np.argmax(np.where(np.arange(V) < V - 50, lg2, -np.inf)))
    big = lg.copy(); big[top[0]] = abs(big[top[0]]) + 1
    assert oq.sample_token(big, 0.0, 1.0, 0, 100.0, [int(top[0])], None, None, 0.0, 1, 0, 0) != int(top[0])
    # top-k: samples only from the k best; EOS restored even when outside the top-k
    seen = {oq.sample_token(lg, 1.5, 1.0, 5, 1.0, None, None, None, 0.0, 7, 0, s) for s in range(300)}
    assert seen <= set(int(t) for t in top[:5]) and len(seen) >= 3
    eos = int(top[100])
    hot = lg.copy(); hot[eos] = lg[top[0]] + 2
    seen = {oq.sample_token(hot, 1.0, 1.0, 5, 1.0, None, None, eos, 0.0, 7, 0, s) for s in range(100)}
    assert eos in seen
    # min-p removes everything far below the maximum
    seen = {oq.sample_token(lg, 1.0, 1.0, 0, 1.0, None, None, None, 0.5, 7, 0, s) for s in range(200)}
    assert all(lg[t] >= lg.max() + np.log(0.5) - 0.05 for t in seen)
    # top-p keeps the head of the distribution
    seen = {oq.sample_token(lg, 1.0, 0.5, 0, 1.0, None, None, None, 0.0, 7, 0, s) for s in range(300)}
    p = np.exp(lg - lg.max()); p /= p.sum()
    assert p[list(seen)].min() >= np.sort(p)[::-1][np.searchsorted(np.cumsum(np.sort(p)[::-1]), 0.5) + 1] * 0.9


def test_frame_loop_is_deterministic_and_teacher_forcing_reproduces_logits():
    for cfg in (oq.TINY, oq.TINY_PROJ):
        W = oq.make_synthetic_weights(cfg)
        m = oq.Qwen3TTSOracle(cfg, W)
        text_ids = [5, 6, 7, cfg.tts_pad_token_id, cfg.tts_pad_token_id, cfg.tts_bos_token_id, 9]
        codec_ids = [-1, -1, -1, cfg.codec_nothink_id, cfg.codec_think_bos_id, cfg.codec_think_eos_id, cfg.codec_bos_id]
        params = dict(temperature=0.9, top_p=1.0, top_k=50, repetition_penalty=1.05, min_p=0.0, seed=3)
        a, la = m.generate_row(text_ids, codec_ids, [11, 12, cfg.tts_eos_token_id], params, max_frames=5)
        b, lb = m.generate_row(text_ids, codec_ids, [11, 12, cfg.tts_eos_token_id], params, max_frames=5)
        assert a.shape == (5, cfg.num_code_groups) and np.array_equal(a, b)
        assert a[:, 0].max() < cfg.talker.vocab_size - 1024 and a[:, 1:].max() < cfg.predictor.vocab_size
        c, lc = m.generate_row(text_ids, codec_ids, [11, 12, cfg.tts_eos_token_id], params, max_frames=5, forced_codes=a)
        assert np.array_equal(c, a) and all(np.array_equal(x, y) for x, y in zip(la, lc[:5]))
