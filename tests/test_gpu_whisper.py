"""-m gpu: Whisper encoder / decoder (HIP) vs the oracle (pinned against HF WhisperModel).

Tolerance (stated): bf16 storage at every primitive boundary on both sides, different accumulation orders:
encoder output  max|dev-ref| <= 0.022*max|ref| and rms <= 0.012*rms(ref);  teacher-forced decoder logits
max <= 0.022*max|ref|, rms <= 0.012*rms(ref) (about twice the observed 0.0106 / 0.0056);  greedy tokens agree where the oracle's top-2 margin exceeds the error."""
import numpy as np
import pytest

import mlx_audio_swift_amd as mas
from gpu_util import observe, rms
from oracle import mel as omel
from oracle import whisper as ow

pytestmark = pytest.mark.gpu


def _host_cfg(c: ow.WhisperConfig) -> mas.WhisperConfig:
    return mas.WhisperConfig(**{k: getattr(c, k) for k in mas.WhisperConfig.__dataclass_fields__})


def _pair(cfg, seed=777):
    W = ow.make_synthetic_weights(cfg, seed=seed)
    return W, ow.WhisperOracle(cfg, W, round="bf16"), mas.WhisperModel.from_weights(_host_cfg(cfg), W)


def _feats(B, n_mels, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((B, 3000, n_mels)) * 0.5).astype(np.float32)


def _check(dev, ref, max_tol, rms_tol):
    scale = float(np.abs(ref).max())
    assert observe("max_rel", float(np.abs(dev - ref).max()) / scale, max_tol), (float(np.abs(dev - ref).max()), scale)
    assert observe("rms_rel", rms(dev, ref) / float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))), rms_tol)


@pytest.mark.parametrize("cfg", [ow.TINY, ow.WhisperConfig(vocab_size=700, num_mel_bins=128, d_model=256, encoder_layers=1,
                                                            encoder_attention_heads=2, encoder_ffn_dim=512, decoder_layers=1,
                                                            decoder_attention_heads=2, decoder_ffn_dim=512)],
                         ids=["d64x2-80mel", "d128x2-128mel"])
def test_encoder_and_teacher_forced_decoder_match_oracle(cfg):
    W, oracle, dev = _pair(cfg)
    B = 2
    feats = _feats(B, cfg.num_mel_bins, 1)
    oracle.reset(B)
    enc_ref = oracle.encode(feats)
    enc = dev.encode(feats)
    for b in range(B):
        _check(enc[b], enc_ref[b].numpy(), 0.022, 0.012)
    rng = np.random.default_rng(2)
    toks = rng.integers(0, cfg.vocab_size, (B, 40))           # > 32 self keys: two key tiles
    dev.decoder_reset()
    got = [dev.decoder_forward(toks[:, t]) for t in range(40)]
    ref = oracle.decode([toks[0], toks[1]])
    for b in range(B):
        d = np.stack([g[b] for g in got]); r = ref[b].numpy()
        _check(d, r, 0.022, 0.012)
        err = float(np.abs(d - r).max())
        top2 = np.sort(r, axis=1)[:, -2:]
        sure = (top2[:, 1] - top2[:, 0]) > 2 * err
        assert sure.sum() > 0 and np.array_equal(d.argmax(1)[sure], r.argmax(1)[sure])


def test_device_synthetic_equals_oracle_weights_and_batch_invariance():
    cfg = ow.TINY
    W, oracle, dev = _pair(cfg, seed=777)
    syn = mas.WhisperModel.synthetic(_host_cfg(cfg), seed=777)
    feats = _feats(3, 80, 3)
    a = dev.encode(feats); b = syn.encode(feats)
    assert np.array_equal(a, b)                                 # same generator => same weights => same numbers
    one = dev.encode(feats[1:2])
    assert np.array_equal(one[0], a[1])                         # row of a batch == the B=1 run
    dev.encode(feats)
    l3 = dev.decoder_forward(np.asarray([5, 6, 7], np.int32))
    dev.encode(feats[2:3])
    l1 = dev.decoder_forward(np.asarray([7], np.int32))
    assert np.array_equal(l1[0], l3[2])


def test_generate_greedy_with_suppress_masks():
    cfg = ow.TINY
    W, oracle, dev = _pair(cfg)
    rng = np.random.default_rng(4)
    t = np.arange(16000 * 4) / 16000.0
    wins = [(0.2 * np.sin(2 * np.pi * 330 * t) + 0.02 * rng.standard_normal(len(t))).astype(np.float32),
            (0.1 * rng.standard_normal(16000 * 2)).astype(np.float32)]
    prompt = [590, 591, 592, 593]
    eot, ts_begin = 599, 560
    gp = mas.STTGenerateParameters(max_tokens=12, temperature=0.0, eot_id=eot, timestamp_begin=ts_begin,
                                   suppress_tokens=[1, 2, 3], begin_suppress_tokens=[eot, 10])
    ids = dev.transcribe_windows(wins, prompt, gp)
    assert len(ids) == 2 and all(len(x) <= 12 for x in ids)
    # replay through the oracle under teacher forcing: the engine's token is the masked argmax up to the logit tolerance
    oracle.reset(2)
    feats = [omel.encoder_features(w, 80)[0] for w in wins]
    oracle.encode(feats)
    for b in range(2):
        seq = prompt + ids[b]
        import torch
        with torch.no_grad():
            lg = oracle.decode_row(b, seq).numpy()
        tol = 0.05 * float(np.abs(lg).max())
        for i, tok in enumerate(ids[b]):
            l = ow.apply_suppress(lg[len(prompt) - 1 + i], i, [eot, 10], [1, 2, 3], ts_begin)
            assert tok < ts_begin and tok not in (1, 2, 3) and (i > 0 or tok not in (eot, 10))
            assert l[tok] >= l.max() - tol, (b, i)
    # deterministic
    assert dev.transcribe_windows(wins, prompt, gp) == ids
    # generate(): chunking + STTOutput bookkeeping (no tokenizer: ids only)
    out = dev.generate(np.concatenate([wins[0]] * 9), gp, prompt_ids=prompt)      # 36 s -> 2 windows
    assert len(out.token_ids) == 2 and out.prompt_tokens == 8 and out.generation_tokens == sum(map(len, out.token_ids))


def test_errors():
    cfg = ow.TINY
    m = mas.WhisperModel(_host_cfg(cfg))
    with pytest.raises(mas.AudioGenerationError) as e:
        m.finalize()
    assert e.value.case == "modelNotInitialized"
    with pytest.raises(mas.AudioGenerationError):
        mas.WhisperModel(mas.WhisperConfig(d_model=100))


def test_mlx_whisper_key_layout_loads_to_the_same_model():
    # WhisperModel.sanitize: the mlx-whisper layout ("encoder.blocks.N.attn.query.weight", conv weights [out, k, in], no encoder
    # positional embedding) must give the model the HF layout gives (WhisperModel.swift:321-478)
    import torch
    cfg = ow.TINY
    W = ow.make_synthetic_weights(cfg, seed=31)
    d = cfg.d_model
    half = d // 2
    inc = np.log(10000.0) / max(half - 1, 1)
    pos = np.arange(1500)[:, None] * np.exp(-inc * np.arange(half))[None]
    W["model.encoder.embed_positions.weight"] = torch.from_numpy(np.concatenate([np.sin(pos), np.cos(pos)], 1).astype(np.float32)).bfloat16()
    hf = mas.WhisperModel.from_weights(_host_cfg(cfg), W)
    attn = {"q_proj": "query", "k_proj": "key", "v_proj": "value", "out_proj": "out"}
    M = {}
    for k, v in W.items():
        k2 = k[len("model."):]
        if k2 == "encoder.embed_positions.weight":
            continue                                                     # omitted by mlx-whisper
        if k2 == "decoder.embed_positions.weight":
            M["decoder.positional_embedding"] = v; continue
        if k2.startswith("decoder.embed_tokens."):
            M["decoder.token_embedding." + k2.split(".", 2)[2]] = v; continue
        if k2 in ("encoder.conv1.weight", "encoder.conv2.weight"):
            M[k2] = v.permute(0, 2, 1).contiguous(); continue             # [out, in, k] -> MLX [out, k, in]
        if k2.startswith("encoder.conv"):
            M[k2] = v; continue
        if k2.startswith("encoder.layer_norm."):
            M["encoder.ln_post." + k2.split(".", 2)[2]] = v; continue
        if k2.startswith("decoder.layer_norm."):
            M["decoder.ln." + k2.split(".", 2)[2]] = v; continue
        stem, _, idx, rest = k2.split(".", 3)
        head, tail = rest.split(".", 1)
        if head == "self_attn_layer_norm":
            r = "attn_ln." + tail
        elif head == "encoder_attn_layer_norm":
            r = "cross_attn_ln." + tail
        elif head == "final_layer_norm":
            r = "mlp_ln." + tail
        elif head in ("fc1", "fc2"):
            r = ("mlp1." if head == "fc1" else "mlp2.") + tail
        else:
            proj, t2 = tail.split(".", 1)
            r = ("attn." if head == "self_attn" else "cross_attn.") + attn[proj] + "." + t2
        M[f"{stem}.blocks.{idx}.{r}"] = v
    M["alignment_heads"] = torch.zeros(2, 2)
    mlx = mas.WhisperModel.from_weights(_host_cfg(cfg), M)
    f = _feats(2, cfg.num_mel_bins, 3)
    a, b = hf.encode(f), mlx.encode(f)
    assert np.array_equal(a, b)
    toks = np.asarray([3, 5], np.int32)
    hf.decoder_reset(); mlx.decoder_reset()
    assert np.array_equal(hf.decoder_forward(toks), mlx.decoder_forward(toks))


def test_generate_stream_tokens_then_result_and_mlx_directory_loader(tmp_path):
    # generateStream (WhisperModel.swift:92-160): token events while the loop runs, ids equal to generate()'s; text deltas are the
    # host's decode-and-diff (:242-254).  Also the directory loader on an mlx-whisper key layout (fromDirectory + sanitize).
    import json
    import torch
    from safetensors.torch import save_file
    cfg = ow.TINY
    W = ow.make_synthetic_weights(cfg, seed=777)
    dev = mas.WhisperModel.from_weights(_host_cfg(cfg), W)
    rng = np.random.default_rng(9)
    wins = [(0.1 * rng.standard_normal(16000 * 3)).astype(np.float32), (0.05 * rng.standard_normal(16000)).astype(np.float32)]
    prompt = [590, 591, 592, 593]
    gp = mas.STTGenerateParameters(max_tokens=14, temperature=0.0, eot_id=599, timestamp_begin=560, suppress_tokens=[1, 2, 3])
    ids = dev.transcribe_windows(wins, prompt, gp)
    ev = list(dev.transcribe_windows_stream(wins, prompt, gp))
    for row in (0, 1):
        toks = [e.token for e in ev if e.row == row and isinstance(e, mas.TokenEvent)]
        assert toks == ids[row] and 599 not in toks
        info = [e.info for e in ev if e.row == row and isinstance(e, mas.InfoEvent)]
        assert len(info) == 1 and info[0].prompt_token_count == 4 and info[0].generation_token_count == len(ids[row])
    assert max(i for i, e in enumerate(ev) if isinstance(e, mas.TokenEvent)) < min(i for i, e in enumerate(ev) if isinstance(e, mas.InfoEvent))

    class Tok:                                             # stand-in tokenizer: an id is one letter; ids the reference would read from it
        end_of_text_id, timestamp_begin_id, is_multilingual = 599, 560, True
        language_to_id = {"en": 591, "fr": 594}

        def decode(self, ids):
            return "".join(chr(97 + (i % 26)) for i in ids)

        def build_prompt_tokens(self, language=None, task="transcribe"):
            return prompt

    dev.tokenizer = Tok()
    gp2 = mas.STTGenerateParameters(max_tokens=14, temperature=0.0, suppress_tokens=[1, 2, 3])     # ids come from the tokenizer
    out = dev.generate(wins[0], gp2)
    assert out.token_ids == [ids[0]] and out.language == "en" and out.text == Tok().decode(ids[0])
    stream = list(dev.generate_stream(wins[0], gp2))
    assert [k for k, _ in stream] == ["token"] * len(ids[0]) + ["result"]
    assert "".join(v for k, v in stream if k == "token") == out.text and stream[-1][1].text == out.text
    dev.tokenizer = None
    with pytest.raises(mas.AudioGenerationError):
        dev.generate(wins[0], mas.STTGenerateParameters(max_tokens=4), prompt_ids=prompt)              # no tokenizer, no ids: loud
    # ---- fromDirectory on the mlx-whisper layout (the guard that rejected ".blocks." keys is gone)
    attn = {"q_proj": "query", "k_proj": "key", "v_proj": "value", "out_proj": "out"}
    M = {}
    for k, v in W.items():
        k2 = k[len("model."):]
        if k2 == "encoder.embed_positions.weight":
            continue
        if k2 == "decoder.embed_positions.weight":
            M["decoder.positional_embedding"] = v; continue
        if k2.startswith("decoder.embed_tokens."):
            M["decoder.token_embedding." + k2.split(".", 2)[2]] = v; continue
        if k2 in ("encoder.conv1.weight", "encoder.conv2.weight"):
            M[k2] = v.permute(0, 2, 1).contiguous(); continue
        if k2.startswith("encoder.conv"):
            M[k2] = v; continue
        if k2.startswith("encoder.layer_norm."):
            M["encoder.ln_post." + k2.split(".", 2)[2]] = v; continue
        if k2.startswith("decoder.layer_norm."):
            M["decoder.ln." + k2.split(".", 2)[2]] = v; continue
        stem, _, idx, rest = k2.split(".", 3)
        head, tail = rest.split(".", 1)
        if head == "self_attn_layer_norm":
            r = "attn_ln." + tail
        elif head == "encoder_attn_layer_norm":
            r = "cross_attn_ln." + tail
        elif head == "final_layer_norm":
            r = "mlp_ln." + tail
        elif head in ("fc1", "fc2"):
            r = ("mlp1." if head == "fc1" else "mlp2.") + tail
        else:
            proj, t2 = tail.split(".", 1)
            r = ("attn." if head == "self_attn" else "cross_attn.") + attn[proj] + "." + t2
        M[f"{stem}.blocks.{idx}.{r}"] = v
    d = tmp_path / "whisper-mlx"
    d.mkdir()
    save_file({k: v.contiguous() for k, v in M.items()}, str(d / "weights.safetensors"))
    hc = _host_cfg(cfg)
    (d / "config.json").write_text(json.dumps({"n_vocab": hc.vocab_size, "n_mels": hc.num_mel_bins, "n_audio_state": hc.d_model,
                                               "n_audio_layer": hc.encoder_layers, "n_audio_head": hc.encoder_attention_heads,
                                               "n_audio_ctx": 1500, "n_text_layer": hc.decoder_layers,
                                               "n_text_head": hc.decoder_attention_heads, "n_text_ctx": hc.max_target_positions,
                                               "encoder_ffn_dim": hc.encoder_ffn_dim, "decoder_ffn_dim": hc.decoder_ffn_dim}))
    (d / "generation_config.json").write_text(json.dumps({"suppress_tokens": [1, 2, 3], "begin_suppress_tokens": [599]}))
    loaded = mas.WhisperModel.from_model_directory(str(d))
    assert loaded.generation_config["suppress_tokens"] == [1, 2, 3]
    # the mlx layout synthesises the encoder sinusoid; compare against the HF-layout model with that same table
    half = cfg.d_model // 2
    inc = np.log(10000.0) / max(half - 1, 1)
    pos = np.arange(1500)[:, None] * np.exp(-inc * np.arange(half))[None]
    W2 = dict(W)
    W2["model.encoder.embed_positions.weight"] = torch.from_numpy(np.concatenate([np.sin(pos), np.cos(pos)], 1).astype(np.float32)).bfloat16()
    hf = mas.WhisperModel.from_weights(hc, W2)
    gp3 = mas.STTGenerateParameters(max_tokens=10, temperature=0.0, eot_id=599, timestamp_begin=560, suppress_tokens=[1, 2, 3])
    assert loaded.transcribe_windows(wins, prompt, gp3) == hf.transcribe_windows(wins, prompt, gp3)


def test_group_of_two_logical_shards_returns_the_unsharded_tokens():
    """mis_whisper_group_generate (BASELINE configs[3]: 30 s chunks sharded over the GPUs of a node): two replicas with the same
    weights on ONE GPU stand in for two devices; five windows split 3 + 2; greedy and sampled (RNG keyed by the global window)."""
    cfg = ow.TINY
    a = mas.WhisperModel.synthetic(_host_cfg(cfg), seed=777)
    b = mas.WhisperModel.synthetic(_host_cfg(cfg), seed=777)
    rng = np.random.default_rng(14)
    wins = [(0.1 * rng.standard_normal(16000 * n)).astype(np.float32) for n in (3, 1, 2, 4, 2)]
    prompt = [590, 591, 592, 593]
    for temp in (0.0, 0.8):
        gp = mas.STTGenerateParameters(max_tokens=10, temperature=temp, seed=5, eot_id=599, timestamp_begin=560)
        whole = a.transcribe_windows(wins, prompt, gp)
        sharded = a.transcribe_windows(wins, prompt, gp, replicas=[a, b])
        assert sharded == whole, temp
    # fewer rows than replicas (the tail slice of a long request): the group runs them on its first `rows` replicas
    assert a.transcribe_windows(wins[:1], prompt, gp, replicas=[a, b]) == a.transcribe_windows(wins[:1], prompt, gp)


def test_encoder_256x256_tile_gemm_is_bit_identical_to_the_128x128_kernel(monkeypatch):
    """k_gemm_big3 (256 x 256 x 64 tiles, eight waves, counted waits; picked when its grid fills the chip) accumulates every output in the
    same k order as k_gemm_big2, so forcing it on every GEMM of the encoder (MIS_GEMM_BIG3=2: conv stem with GELU / GELU + positions,
    qkv, out_proj + residual, fc1 + GELU, fc2 + residual; ragged M = 2 x 1500 and N = 256 / 768 / 512 against 256-wide tiles) must
    reproduce the 128 x 128 kernel's encoder output bit for bit - and with it the oracle bound of the test above."""
    cfg = ow.WhisperConfig(vocab_size=700, num_mel_bins=128, d_model=256, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=512,
                           decoder_layers=1, decoder_attention_heads=2, decoder_ffn_dim=512)
    W, oracle, dev = _pair(cfg)
    feats = _feats(2, cfg.num_mel_bins, 4)
    monkeypatch.setenv("MIS_GEMM_BIG3", "0")
    small = dev.encode(feats)
    monkeypatch.setenv("MIS_GEMM_BIG3", "2")
    big = dev.encode(feats)
    assert np.array_equal(small, big)
    oracle.reset(2)
    ref = oracle.encode(feats)
    for b in range(2):
        _check(big[b], ref[b].numpy(), 0.022, 0.012)


def test_encoder_attention_on_256_row_blocks_is_bit_identical_to_the_16_rows_per_wave_kernel(monkeypatch):
    """`k_attn_prefill2` (round 6: 256 query rows per block, the K/V tile staged once per block through LDS, two row groups per wave, mask on
    the last tile only, rescale only when a row maximum moved) performs the same operations per query row in the same order as
    `k_attn_prefill` (`MIS_ATTN_PREFILL_V1=1`): encoder outputs must be equal bit for bit - 1500 positions = 5 full blocks + 220 rows,
    47 key tiles of which the last is masked."""
    cfg = ow.WhisperConfig(vocab_size=700, num_mel_bins=128, d_model=256, encoder_layers=2, encoder_attention_heads=4, encoder_ffn_dim=512,
                           decoder_layers=1, decoder_attention_heads=4, decoder_ffn_dim=512)
    dev = mas.WhisperModel.synthetic(_host_cfg(cfg), seed=31)
    feats = _feats(3, cfg.num_mel_bins, 9)
    got = {}
    for v1 in ("1", "0"):
        monkeypatch.setenv("MIS_ATTN_PREFILL_V1", v1)
        got[v1] = dev.encode(feats)
    monkeypatch.delenv("MIS_ATTN_PREFILL_V1")
    assert np.isfinite(got["0"]).all() and np.abs(got["0"]).max() > 0
    assert np.array_equal(got["0"], got["1"])


def test_cross_attention_with_two_pairs_of_key_tiles_in_flight_is_bit_identical_to_the_pair_at_a_time_loop(monkeypatch):
    """`k_attn_decode<64, 2, XS>` (round 4: the schedule of the cross-attention over the 1500 encoder positions - 47 key tiles, five or six per
    wave, two pairs of them in registers, non-temporal loads) multiplies every wave's tiles in the order of the loop it replaces
    (`MIS_ATTN_XS=0`): teacher-forced decoder logits must be equal bit for bit, every head, every step."""
    cfg = ow.WhisperConfig(vocab_size=700, num_mel_bins=128, d_model=256, encoder_layers=1, encoder_attention_heads=4, encoder_ffn_dim=512,
                           decoder_layers=2, decoder_attention_heads=4, decoder_ffn_dim=512)
    W, oracle, dev = _pair(cfg)
    B = 3
    dev.encode(_feats(B, cfg.num_mel_bins, 6))
    toks = np.random.default_rng(8).integers(0, cfg.vocab_size, (B, 6))
    got = {}
    for xs in ("0", "1"):
        monkeypatch.setenv("MIS_ATTN_XS", xs)
        dev.decoder_reset()
        got[xs] = np.stack([dev.decoder_forward(toks[:, t]) for t in range(toks.shape[1])])
    monkeypatch.delenv("MIS_ATTN_XS")
    assert np.isfinite(got["1"]).all() and np.abs(got["1"]).max() > 0
    assert np.array_equal(got["0"], got["1"])


def test_sampler_timeout_inside_transcribe_recovers(monkeypatch):
    """A vocabulary wider than 4 096 ids below the timestamp range takes the one-launch full-vocabulary sampler (k_samp_cluster: 8 blocks
    per window that wait for each other).  MIS_SAMPLER_SPIN=0 makes every one of its row barriers time out at once: the loop sees the
    flag at its first poll - before any id of that interval is announced -, resets the decoder and runs prompt + loop again on the
    multi-launch sampler; ids, the stream's token events and the per-window counts equal the undisturbed run's
    (WhisperModel.swift:186-282 has no such failure mode, so the engine must not surface one)."""
    lib = mas._lib.lib()
    cfg = ow.WhisperConfig(vocab_size=6000, num_mel_bins=80, d_model=128, encoder_layers=1, encoder_attention_heads=2, encoder_ffn_dim=256,
                           decoder_layers=2, decoder_attention_heads=2, decoder_ffn_dim=256)
    dev = mas.WhisperModel.synthetic(_host_cfg(cfg), seed=99)
    rng = np.random.default_rng(8)
    wins = [(0.1 * rng.standard_normal(16000 * n)).astype(np.float32) for n in (3, 1, 2)]
    prompt = [5990, 5991, 5992, 5993]
    for temp in (0.0, 0.7):
        gp = mas.STTGenerateParameters(max_tokens=20, temperature=temp, seed=5, eot_id=5999, timestamp_begin=5900)
        monkeypatch.delenv("MIS_SAMPLER_SPIN", raising=False)
        want = dev.transcribe_windows(wins, prompt, gp)
        before = lib.mis_debug_sampler_failures()
        monkeypatch.setenv("MIS_SAMPLER_SPIN", "0")
        assert dev.transcribe_windows(wins, prompt, gp) == want, temp
        assert lib.mis_debug_sampler_failures() == before + 1
        seen = [[] for _ in wins]
        for ev in dev.transcribe_windows_stream(wins, prompt, gp):
            if isinstance(ev, mas.TokenEvent):
                seen[ev.row].append(ev.token)
        assert seen == [[t for t in w if t != 5999] for w in want], temp
        monkeypatch.setenv("MIS_SAMPLER_WIDE", "1")                    # the multi-launch path from the start: nothing to recover from
        assert dev.transcribe_windows(wins, prompt, gp) == want and lib.mis_debug_sampler_failures() == before + 2
        monkeypatch.delenv("MIS_SAMPLER_WIDE")


@pytest.mark.parametrize("B", [3, 8, 11, 17])
def test_glue_folded_into_its_consumers_matches_the_separate_launches_and_the_oracle(monkeypatch, B):
    """Round 6: up to 16 rows two of the three residual + LayerNorm launches of a decoder layer run in the prologue of their consumer
    (`MIS_WHISPER_FOLD`, a bit mask): bit 4 = LayerNorm 2 AND the cross-attention's query projection inside the cross-attention kernel
    (`k_attn_decode<64, 2, true, QP>`: a block is one (row, head) and rebuilds only its row), bit 2 = LayerNorm 3 inside fc1's prologue
    (`k_gemm_skinny_norm`: every block rebuilds all rows from the producer's split-K slabs and the residual stream); both are the
    default.  The residual stream alternates between two buffers.  At large-v3's decoder WIDTH (d 1280, 20 heads, ffn 5120; three layers,
    small vocabulary): every form is within the stated tolerance of the oracle and no further from the separate launches (`=0`) than
    that (same arithmetic per element; row statistics and the query projection's K sum are float32 sums in another order).  B = 11: rows
    8 .. 10 are a wave's SECOND row in fc1's prologue; B = 17: two m-tiles - the engine keeps the separate launches (bit 3: fail where a
    requested fold does not apply)."""
    cfg = ow.WhisperConfig(vocab_size=700, num_mel_bins=128, d_model=1280, encoder_layers=1, encoder_attention_heads=20, encoder_ffn_dim=1280,
                           decoder_layers=3, decoder_attention_heads=20, decoder_ffn_dim=5120)
    W, oracle, dev = _pair(cfg)
    feats = _feats(B, cfg.num_mel_bins, 16)
    dev.encode(feats)
    T = 5
    toks = np.random.default_rng(18).integers(0, cfg.vocab_size, (B, T))
    strict = B <= 16
    modes = {"separate": "0", "fc1_fold": "12" if strict else "4", "attention_fold": "24" if strict else "16", "both": "28" if strict else "20",
             "default": None}
    got = {}
    for name, fold in modes.items():
        if fold is None:
            monkeypatch.delenv("MIS_WHISPER_FOLD", raising=False)
        else:
            monkeypatch.setenv("MIS_WHISPER_FOLD", fold)
        dev.decoder_reset()
        got[name] = np.stack([dev.decoder_forward(toks[:, t]) for t in range(T)], axis=1)        # [B, T, V]
    if not strict:
        for m in ("12", "24"):
            monkeypatch.setenv("MIS_WHISPER_FOLD", m)
            with pytest.raises(mas.AudioGenerationError):
                dev.decoder_forward(toks[:, 0])
    monkeypatch.delenv("MIS_WHISPER_FOLD", raising=False)
    rows = sorted({0, B // 2, B - 1, min(8, B - 1)})
    oracle.reset(B)
    oracle.encode(feats)
    ref = oracle.decode([toks[b] for b in range(B)])
    for b in rows:
        r = ref[b].numpy()
        for name in modes:
            _check(got[name][b], r, 0.022, 0.012)
    from gpu_util import record
    scale = float(np.abs(got["separate"]).max())
    for name in ("fc1_fold", "attention_fold", "both", "default"):
        d_max = float(np.abs(got[name] - got["separate"]).max()) / scale
        d_rms = rms(got[name], got["separate"]) / float(np.sqrt(np.mean(got["separate"].astype(np.float64) ** 2)))
        record(f"whisper_decoder_glue_fold_b{B}_{name}", vs_separate_max_rel=d_max, vs_separate_rms_rel=d_rms, tol_max=0.022, tol_rms=0.012)
        if not strict:
            assert np.array_equal(got[name], got["separate"])
        else:
            assert d_max <= 0.022 and d_rms <= 0.012, (name, d_max, d_rms)
    if strict:
        assert np.array_equal(got["default"], got["both"])                                        # (the default IS bits 2 and 4)
