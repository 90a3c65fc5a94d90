"""-m gpu: the batched generate()/generateStream() path end to end: prefill -> graph-replayed decode loop
with on-device sampling -> parseOutput -> de-interleave -> SNAC, checked against the oracles.

Free-running tokens cannot be compared one by one (a one-ulp bf16 difference flips an argmax and the
sequences diverge), so parity is stated as: (1) every token the engine emitted, replayed through the
oracle LM under teacher forcing, is the oracle's argmax up to the stated logit tolerance; (2) the
waveform equals the oracle SNAC decode of the engine's own tokens within 1e-4 RMS; (3) integer framing
is exact."""
import ctypes as C

import numpy as np
import pytest

import mlx_audio_swift_amd as mas
from gpu_util import lm_pair, rms, snac_pair
from oracle import llama as ollama
from oracle import orpheus_codes as oc
from oracle import snac as osnac

pytestmark = pytest.mark.gpu

SNAC_SMALL = dict(encoder_dim=4, encoder_rates=[2, 4, 8, 8], decoder_dim=64, decoder_rates=[8, 8, 4, 2],
                  codebook_size=4096, codebook_dim=8, vq_strides=[4, 2, 1])          # hop 512 like snac_24khz
LM_SMALL = ollama.LlamaConfig(hidden_size=256, num_hidden_layers=2, intermediate_size=512, num_attention_heads=2,
                              num_key_value_heads=1, head_dim=128, vocab_size=156940)


@pytest.fixture(scope="module")
def stack():
    ocfg_s, osn, dsn = snac_pair(SNAC_SMALL)
    W, olm, dlm = lm_pair(LM_SMALL, codec=dsn)
    return ocfg_s, osn, dsn, olm, dlm


def _prompts(rng, lens):
    return [np.asarray([oc.START_OF_HUMAN] + list(rng.integers(0, 128000, n - 4)) +
                       [oc.END_OF_TEXT, oc.END_OF_HUMAN, oc.START_OF_SPEECH], np.int32) for n in lens]


def test_generate_greedy_constrained_parity(stack):
    ocfg_s, osn, dsn, olm, dlm = stack
    rng = np.random.default_rng(0)
    prompts = _prompts(rng, [9, 5, 12])
    gp = mas.GenerateParameters(max_tokens=21, temperature=0.0, top_p=0.8, repetition_penalty=1.3, seed=3,
                                frame_constrained=True)
    zeros = [np.zeros((3, n), np.float32) for n in dsn.noise_lengths(3)]
    pcm, toks = dlm.generate_batch(prompts, gp, snac_noise=zeros, return_tokens=True)
    assert [len(t) for t in toks] == [21, 21, 21] and [len(p) for p in pcm] == [3 * 2048] * 3
    # (3) framing exact; (2) waveform vs oracle decode of the engine's tokens
    for b in range(3):
        slots = (toks[b] - oc.AUDIO_TOKEN_OFFSET) // 4096
        assert np.array_equal(slots, np.arange(21) % 7)
        codes = oc.parse_output_row(list(prompts[b]) + list(toks[b]))
        assert np.array_equal(codes, toks[b] - oc.AUDIO_TOKEN_OFFSET)
        l0, l1, l2 = oc.deinterleave(codes)
        ref = osn.decode([l0[None], l1[None], l2[None]], None)[0, 0]
        assert rms(pcm[b], ref) < 1e-4
    # (1) teacher-forced oracle: engine's token is the oracle argmax (after the oracle's own penalty) within tolerance
    from oracle import sampler as osamp
    olm.reset(3)
    for b in range(3):
        seq = list(prompts[b]) + list(toks[b])
        logits = olm._forward_row(b, __import__("torch").as_tensor(np.asarray(seq[:-1], np.int64))).numpy()
        win = osamp.RepetitionWindow(20, prompts[b])
        for i, t in enumerate(toks[b]):
            l = osamp.apply_repetition_penalty(logits[len(prompts[b]) - 1 + i], win.ids, 1.3, bf16=True)
            lo = oc.AUDIO_TOKEN_OFFSET + (i % 7) * 4096
            seg = l[lo: lo + 4096]
            tol = 0.04 * float(np.abs(logits).max())
            assert seg[t - lo] >= seg.max() - tol, (b, i)
            win.push(int(t))
    # determinism + batch-vs-single parity on the token level
    pcm2, toks2 = dlm.generate_batch(prompts, gp, snac_noise=zeros, return_tokens=True)
    assert all(np.array_equal(a, b) for a, b in zip(toks, toks2)) and all(np.array_equal(a, b) for a, b in zip(pcm, pcm2))
    one_pcm, one_tok = dlm.generate_batch(prompts[1:2], gp, snac_noise=[z[:1] for z in zeros], return_tokens=True)
    assert np.array_equal(one_tok[0], toks[1]) and np.array_equal(one_pcm[0], pcm[1])


def test_sampled_generation_is_seeded_and_stream_contract(stack):
    ocfg_s, osn, dsn, olm, dlm = stack
    rng = np.random.default_rng(1)
    prompts = _prompts(rng, [8, 8])
    gp = mas.GenerateParameters(max_tokens=14, temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=11,
                                frame_constrained=True)
    a = dlm.generate_batch(prompts, gp, return_tokens=True)
    b = dlm.generate_batch(prompts, gp, return_tokens=True)
    gp2 = mas.GenerateParameters(max_tokens=14, temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=12,
                                 frame_constrained=True)
    c = dlm.generate_batch(prompts, gp2, return_tokens=True)
    assert all(np.array_equal(x, y) for x, y in zip(a[1], b[1])) and all(np.array_equal(x, y) for x, y in zip(a[0], b[0]))
    assert not all(np.array_equal(x, y) for x, y in zip(a[1], c[1]))
    # row_offset makes a shard reproduce the rows of the full batch (sharding invariance)
    gp_shard = mas.GenerateParameters(max_tokens=14, temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=11,
                                      frame_constrained=True, row_offset=1)
    s = dlm.generate_batch(prompts[1:], gp_shard, return_tokens=True)
    assert np.array_equal(s[1][0], a[1][1]) and np.array_equal(s[0][0], a[0][1])
    # stream contract (LlamaTTS.swift:862,893-904): .token* then .info then ONE .audio per row
    ev = list(dlm.generate_stream_batch(prompts, gp))
    for row in (0, 1):
        kinds = [type(e).__name__ for e in ev if e.row == row]
        assert kinds == ["TokenEvent"] * 14 + ["InfoEvent", "AudioEvent"]
        toks = [e.token for e in ev if e.row == row and isinstance(e, mas.TokenEvent)]
        assert np.array_equal(toks, a[1][row])
        audio = [e.audio for e in ev if e.row == row and isinstance(e, mas.AudioEvent)][0]
        assert np.array_equal(audio, a[0][row])
        info = [e.info for e in ev if e.row == row and isinstance(e, mas.InfoEvent)][0]
        assert info.prompt_token_count == 8 and info.generation_token_count == 14 and info.tokens_per_second > 0


def test_eos_ragged_rows_errors_and_cancel(stack):
    ocfg_s, osn, dsn, olm, dlm = stack
    rng = np.random.default_rng(2)
    prompts = _prompts(rng, [6, 6])
    # unconstrained greedy on random weights: tokens are arbitrary ids -> almost surely < 7 audio tokens
    gp = mas.GenerateParameters(max_tokens=5, temperature=0.0, repetition_penalty=0.0, seed=1)
    with pytest.raises(mas.AudioGenerationError) as e:
        dlm.generate_batch(prompts, gp)
    assert e.value.case == "generationFailed"                   # "No audio codes generated", LlamaTTS.swift:754-756
    with pytest.raises(mas.AudioGenerationError) as e:
        dlm.generate_batch([np.asarray([200000], np.int32)], gp)
    assert e.value.case == "invalidInput"
    flag = C.c_int(1)
    gpc = mas.GenerateParameters(max_tokens=64, temperature=0.0, frame_constrained=True)
    with pytest.raises(mas.AudioGenerationError) as e:
        list(dlm.generate_stream_batch(prompts, gpc, cancel_flag=flag))
    assert e.value.case == "cancelled"
    no_codec = mas.LlamaTTSModel.synthetic(mas.LlamaTTSConfiguration(
        hidden_size=64, num_hidden_layers=1, intermediate_size=64, num_attention_heads=1, num_key_value_heads=1,
        head_dim=64, vocab_size=200))
    with pytest.raises(mas.AudioGenerationError) as e:
        no_codec.generate_batch([np.asarray([1, 2], np.int32)], gp)
    assert e.value.case == "modelNotInitialized"                # "SNAC model not loaded", LlamaTTS.swift:672-674


def test_full_size_orpheus_3b_properties():
    """BASELINE configs[2] at its real size (Orpheus-3B dims, snac_24khz, batch 32, synthetic weights) through size-independent
    properties: seeded determinism, shard / batch-width invariance of tokens AND waveform (a row computed in a 32-row batch equals
    the same row computed alone with its global row offset), exact integer framing, waveform range."""
    from mlx_audio_swift_amd.synthetic import snac_synthetic_weights
    snac_cfg = mas.SNACConfig()
    codec = mas.SNAC.from_weights(snac_cfg, snac_synthetic_weights(snac_cfg, seed=1234))
    cfg = mas.LlamaTTSConfiguration(rope_theta=500000.0, rope_scaling={"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                                                    "original_max_position_embeddings": 8192, "rope_type": "llama3"})
    lm = mas.LlamaTTSModel.synthetic(cfg, codec=codec, seed=4321)
    rng = np.random.default_rng(9)
    prompts = _prompts(rng, [32] * 32)
    gp = mas.GenerateParameters(max_tokens=21, temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=2024, frame_constrained=True)
    pcm, toks = lm.generate_batch(prompts, gp, return_tokens=True)
    pcm2, toks2 = lm.generate_batch(prompts, gp, return_tokens=True)
    hop = 2048
    for r in range(32):
        assert np.array_equal(toks[r], toks2[r]) and np.array_equal(pcm[r], pcm2[r])
        assert len(toks[r]) == 21 and len(pcm[r]) == 3 * hop
        codes = oc.parse_output_row(np.concatenate([prompts[r], toks[r]]))
        assert len(codes) == 21 and codes.min() >= 0 and codes.max() < 7 * 4096
        assert np.isfinite(pcm[r]).all() and np.abs(pcm[r]).max() <= 1.0
    for r in (0, 17, 31):
        gp1 = mas.GenerateParameters(max_tokens=21, temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=2024, frame_constrained=True,
                                     row_offset=r)
        p1, t1 = lm.generate_batch([prompts[r]], gp1, return_tokens=True)
        assert np.array_equal(t1[0], toks[r]) and np.array_equal(p1[0], pcm[r])


def test_prompt_without_speech_marker_counts_towards_the_audio(stack):
    """parseOutput keeps the WHOLE sequence when the prompt has no start-of-speech marker (LlamaTTS.swift:749-752 ->
    :383-434), so prompt tokens that are audio codes are decoded too: the PCM bound must include the prompt length."""
    ocfg_s, osn, dsn, olm, dlm = stack
    rng = np.random.default_rng(5)
    frames = 3
    prompt = np.asarray([oc.AUDIO_TOKEN_OFFSET + (j % 7) * 4096 + int(rng.integers(0, 4096)) for j in range(7 * frames)], np.int32)
    gp = mas.GenerateParameters(max_tokens=7, temperature=0.0, repetition_penalty=0.0, frame_constrained=True)
    pcm, toks = dlm.generate_batch([prompt], gp, return_tokens=True)
    assert len(toks[0]) == 7
    codes = oc.parse_output_row(np.concatenate([prompt, toks[0]]))
    assert len(codes) == 7 * (frames + 1)
    hop = int(np.prod(SNAC_SMALL["decoder_rates"])) * SNAC_SMALL["vq_strides"][0]
    assert len(pcm[0]) == (frames + 1) * hop


def test_vyvotts_token_ids_drive_the_same_loop():
    # VyvoTTS = Qwen3-style LM + SNAC with its own token ids (Qwen3.swift:19-29): EOS, frame-constrained range and parse use them
    ocfg_s, osn, dsn = snac_pair(SNAC_SMALL)
    V = mas.VyvoTokens
    lcfg = ollama.LlamaConfig(hidden_size=256, num_hidden_layers=1, intermediate_size=512, num_attention_heads=2, num_key_value_heads=1,
                              head_dim=128, vocab_size=V.audio_token_offset + 7 * 4096, rope_theta=1e6, rope_scaling=None,
                              tie_word_embeddings=True, qk_norm=True, rope_plain=True, rms_norm_eps=1e-6)
    from gpu_util import lm_host_config
    hc = lm_host_config(lcfg)
    hc.start_of_speech_id, hc.end_of_speech_id, hc.audio_token_offset, hc.start_of_ai_id = (V.start_of_speech, V.end_of_speech,
                                                                                           V.audio_token_offset, V.start_of_ai)
    lm = mas.LlamaTTSModel.synthetic(hc, codec=dsn, seed=99)
    prompt = np.asarray([V.start_of_human, 11, 12, V.end_of_text, V.end_of_human, V.start_of_ai, V.start_of_speech], np.int32)
    gp = mas.GenerateParameters(max_tokens=14, temperature=0.0, repetition_penalty=0.0, frame_constrained=True)
    zeros = [np.zeros((1, n), np.float32) for n in dsn.noise_lengths(2)]
    pcm, toks = lm.generate_batch([prompt], gp, snac_noise=zeros, return_tokens=True)
    slots = (toks[0] - V.audio_token_offset) // 4096
    assert len(toks[0]) == 14 and np.array_equal(slots, np.arange(14) % 7) and len(pcm[0]) == 2 * 2048
    l0, l1, l2 = oc.deinterleave(oc.parse_output_row_vyvo(list(prompt) + list(toks[0])))
    ref = osn.decode([l0[None], l1[None], l2[None]], None)[0, 0]
    assert rms(pcm[0], ref) < 1e-4


def test_vyvotts_decodes_snac_in_independent_chunks():
    # decodeAudioFromCodes (Qwen3.swift:47-83): utterances longer than codec_chunk_groups are decoded chunk by chunk, every chunk
    # an independent SNAC decode; explicit noise is sliced at the chunk offsets
    ocfg_s, osn, dsn = snac_pair(SNAC_SMALL)
    V = mas.VyvoTokens
    lcfg = ollama.LlamaConfig(hidden_size=256, num_hidden_layers=1, intermediate_size=512, num_attention_heads=2, num_key_value_heads=1,
                              head_dim=128, vocab_size=V.audio_token_offset + 7 * 4096, rope_theta=1e6, rope_scaling=None,
                              tie_word_embeddings=True, qk_norm=True, rope_plain=True, rms_norm_eps=1e-6)
    from gpu_util import lm_host_config
    hc = lm_host_config(lcfg)
    hc.start_of_speech_id, hc.end_of_speech_id, hc.audio_token_offset, hc.start_of_ai_id = (V.start_of_speech, V.end_of_speech,
                                                                                           V.audio_token_offset, V.start_of_ai)
    hc.codec_chunk_groups = 3
    lm = mas.LlamaTTSModel.synthetic(hc, codec=dsn, seed=99)
    rng = np.random.default_rng(21)
    prompts = [np.asarray([V.start_of_human, 11 + r, 12, V.end_of_text, V.end_of_human, V.start_of_ai, V.start_of_speech], np.int32)
               for r in range(2)]
    groups = 7                                                                  # chunks of 3, 3 and 1 groups
    gp = mas.GenerateParameters(max_tokens=7 * groups, temperature=0.0, repetition_penalty=0.0, frame_constrained=True)
    noise = [rng.standard_normal((2, n)).astype(np.float32) for n in dsn.noise_lengths(groups)]
    pcm, toks = lm.generate_batch(prompts, gp, snac_noise=noise, return_tokens=True)
    whole = []
    for r in range(2):
        assert len(toks[r]) == 7 * groups and len(pcm[r]) == groups * 2048
        codes = oc.parse_output_row_vyvo(list(prompts[r]) + list(toks[r]))
        ref = oc.decode_audio_from_codes_chunked(codes, osn, chunk_groups=3, noises=[z[r:r + 1] for z in noise])
        assert rms(pcm[r], ref) < 1e-4
        whole.append(oc.decode_audio_from_codes_chunked(codes, osn, chunk_groups=50, noises=[z[r:r + 1] for z in noise]))
    assert rms(pcm[0], whole[0]) > 1e-3                                         # NOT the single-decode waveform
    # internal noise: runs, deterministic, and rows of different length (EOS) take the ragged sub-batches
    a = lm.generate_batch(prompts, gp)
    b = lm.generate_batch(prompts, gp)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and len(a[0]) == groups * 2048


def test_device_group_two_logical_shards_equal_the_unsharded_call_bitwise(stack):
    """Multi-GPU behind the C ABI on ONE GPU: a group of two replicas (two LM + codec handles on device 0, each
    with its own stream and worker thread inside mis_tts_group_generate*) must reproduce the single-handle call bit for bit -
    tokens and waveform - because the RNG is keyed by the global row.  Host gather and device all-gather forms."""
    import torch
    from gpu_util import snac_pair
    from mlx_audio_swift_amd.sharding import TTSGroup
    ocfg_s, osn, dsn, olm, dlm = stack
    rng = np.random.default_rng(31)
    prompts = _prompts(rng, [6, 9, 7, 8, 6])
    gp = mas.GenerateParameters(max_tokens=21, temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=17, frame_constrained=True)
    ref_pcm, ref_tok = dlm.generate_batch(prompts, gp, return_tokens=True)
    # two fresh replicas with the same weights (what a host does per device: load the checkpoint once per GPU)
    from gpu_util import lm_pair
    reps = []
    for _ in range(2):
        _, _, codec = snac_pair(SNAC_SMALL)
        _, _, lm = lm_pair(LM_SMALL, codec=codec)
        reps.append(lm)
    grp = TTSGroup(reps)
    assert len(grp) == 2
    pcm, tok = grp.generate_batch(prompts, gp, return_tokens=True)
    for b in range(5):
        assert np.array_equal(tok[b], ref_tok[b]) and np.array_equal(pcm[b], ref_pcm[b]), b
    t = grp.last_timing()
    assert t["n_shards"] == 2 and t["generate_ms"] > 0 and t["slowest_shard_ms"] <= t["generate_ms"] + 1.0
    # device form: both replicas' buffers hold ALL rows afterwards (all-gather by peer copies)
    n = dsn.num_samples(3)
    bufs = [torch.zeros((5, n), dtype=torch.float32, device="cuda:0") for _ in range(2)]
    lens, ntok = grp.generate_device(prompts, gp, [b.data_ptr() for b in bufs], n)
    torch.cuda.synchronize()
    assert lens == [n] * 5 and ntok == [21] * 5
    for buf in bufs:
        got = buf.cpu().numpy()
        for b in range(5):
            assert np.array_equal(got[b], ref_pcm[b]), b
    # a failing shard surfaces as an error naming the shard, and the group stays usable
    bad = [p.copy() for p in prompts]
    bad[4][0] = 10 ** 6
    with pytest.raises(mas.AudioGenerationError) as e:
        grp.generate_batch(bad, gp)
    assert e.value.case == "invalidInput" and "shard 1" in str(e.value)
    pcm2 = grp.generate_batch(prompts, gp)
    assert all(np.array_equal(a, b) for a, b in zip(pcm2, ref_pcm))
    with pytest.raises(mas.AudioGenerationError):
        TTSGroup([reps[0], reps[0]])                       # one in-flight call per handle


def test_rccl_communicator_world1_all_gather_is_the_identity():
    """mis_comm_* (one process per GPU): with a single rank the RCCL all-gather must return the local block - exercises librccl
    loading, ncclCommInitRank from a unique id and ncclAllGather on the library stream (N > 1 needs more GPUs than a test box has)."""
    import torch
    from mlx_audio_swift_amd.sharding import Communicator
    comm = Communicator(0, 0, 1)
    x = torch.arange(3 * 1000, dtype=torch.float32, device="cuda:0").reshape(3, 1000)
    out = torch.zeros_like(x)
    lens, ms = comm.all_gather_pcm(x.data_ptr(), [1000, 7, 0], 3, 1000, out.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out, x) and lens.tolist() == [1000, 7, 0] and ms >= 0.0
    comm.close()


def test_full_vocabulary_sampler_paths_agree_inside_generate(monkeypatch):
    """Real Orpheus decoding is not frame-constrained: every step samples over the whole 156 940-id vocabulary (with random weights
    that yields no frames at all - `No audio codes generated`, like the reference).  frame_constrained = 2 keeps the frame range but
    sends it through the full-vocabulary kernels: the one-launch sampler (k_samp_cluster) replayed inside the step graph - the same
    per-row exchange area step after step - returns the tokens of the six-kernel path (MIS_SAMPLER_WIDE=1) and of the narrow
    single-launch sampler (frame_constrained = 1); all three are pinned on the oracle in test_gpu_sampler.py."""
    cfg = mas.LlamaTTSConfiguration(hidden_size=256, num_hidden_layers=2, intermediate_size=512, num_attention_heads=2,
                                    num_key_value_heads=1, head_dim=128, vocab_size=156940, rope_theta=500000.0)
    snac_cfg = mas.SNACConfig(**SNAC_SMALL)
    from mlx_audio_swift_amd.synthetic import snac_synthetic_weights
    codec = mas.SNAC.from_weights(snac_cfg, snac_synthetic_weights(snac_cfg, seed=1234))
    lm = mas.LlamaTTSModel.synthetic(cfg, codec=codec, seed=77)
    rng = np.random.default_rng(3)
    prompts = _prompts(rng, [9, 14, 6, 11, 8])
    toks = {}
    for name, mode, fc, n in (("cluster", "0", 2, 42), ("six", "1", 2, 49), ("narrow", "0", 1, 56)):
        monkeypatch.setenv("MIS_SAMPLER_WIDE", mode)           # (different token budgets: the step graph is captured again per path)
        gp = mas.GenerateParameters(max_tokens=n, temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=11, frame_constrained=fc)
        _, t = lm.generate_batch(prompts, gp, return_tokens=True)
        toks[name] = t
    for r in range(len(prompts)):
        assert len(toks["cluster"][r]) == 42
        assert np.array_equal(toks["cluster"][r], toks["six"][r][:42]) and np.array_equal(toks["cluster"][r], toks["narrow"][r][:42]), r
    monkeypatch.setenv("MIS_SAMPLER_WIDE", "0")
    with pytest.raises(mas.AudioGenerationError) as e:         # unconstrained + random weights: no frame survives parseOutput (:754-756)
        lm.generate_batch(prompts, mas.GenerateParameters(max_tokens=14, temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=11))
    assert "No audio codes" in str(e.value)


def test_one_launch_sampler_timeout_inside_generate_recovers(monkeypatch):
    """The reference's decode loop cannot fail for lack of free compute units (LlamaTTS.swift:714-744), so neither may the engine's: a
    timed-out row barrier of the one-launch sampler (MIS_SAMPLER_SPIN=0: a block gives up without polling - every step fails) is seen
    at the first poll of the loop, before a single token of that poll interval is announced; the request then runs once more on the
    multi-launch sampler.  The call succeeds, tokens and waveform equal the undisturbed run's, the stream announces every token
    exactly once and in order, the failure was counted once, and the handle keeps working (the switches are part of the graph key: no
    stale graph is replayed)."""
    lib = mas._lib.lib()
    cfg = mas.LlamaTTSConfiguration(hidden_size=256, num_hidden_layers=2, intermediate_size=512, num_attention_heads=2,
                                    num_key_value_heads=1, head_dim=128, vocab_size=156940, rope_theta=500000.0)
    snac_cfg = mas.SNACConfig(**SNAC_SMALL)
    from mlx_audio_swift_amd.synthetic import snac_synthetic_weights
    codec = mas.SNAC.from_weights(snac_cfg, snac_synthetic_weights(snac_cfg, seed=1234))
    lm = mas.LlamaTTSModel.synthetic(cfg, codec=codec, seed=77)
    prompts = _prompts(np.random.default_rng(3), [9, 14, 6, 11, 8, 7, 12, 10])
    gp = mas.GenerateParameters(max_tokens=42, temperature=0.6, top_p=0.8, repetition_penalty=1.3, seed=11, frame_constrained=2)
    monkeypatch.setenv("MIS_SAMPLER_WIDE", "0")
    pcm_want, want = lm.generate_batch(prompts, gp, return_tokens=True)
    before = lib.mis_debug_sampler_failures()
    monkeypatch.setenv("MIS_SAMPLER_SPIN", "0")
    pcm_got, got = lm.generate_batch(prompts, gp, return_tokens=True)          # same parameters, same handle: the key differs by the switch
    assert lib.mis_debug_sampler_failures() == before + 1
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    for a, b in zip(pcm_want, pcm_got):
        assert np.array_equal(a, b)
    # the stream form: token events per row = the row's tokens, once, in order (nothing of the failed attempt leaks out)
    seen = [[] for _ in prompts]
    for ev in lm.generate_stream_batch(prompts, gp):
        if isinstance(ev, mas.TokenEvent):
            seen[ev.row].append(ev.token)
    assert lib.mis_debug_sampler_failures() == before + 2
    for r, row in enumerate(seen):
        assert row == list(want[r]), r
    monkeypatch.delenv("MIS_SAMPLER_SPIN")
    _, again = lm.generate_batch(prompts, gp, return_tokens=True)
    assert lib.mis_debug_sampler_failures() == before + 2
    for a, b in zip(want, again):
        assert np.array_equal(a, b)
