"""CPU: pins for oracle/dac.py: the reference's own shape tests and a naive transposed-conv restatement."""
import numpy as np
import torch

from oracle import dac as od


def test_reference_shape_pins():
    # Tests/MLXAudioCodecsTests.swift:1127-1194
    assert od.num_samples(od.DacConfig(decoder_rates=(8, 5, 4, 2)), 250) == 80_043
    assert od.num_samples(od.DacConfig(decoder_rates=(8, 5, 4, 2)), 375) == 120_043
    assert od.num_samples(od.DacConfig(decoder_rates=(8, 8, 4, 2)), 430) == 220_235


def test_transposed_conv_with_output_padding_matches_naive():
    rng = np.random.default_rng(0)
    cfg = od.TINY
    W = od.make_synthetic_weights(cfg)
    m = od.DacOracle(cfg, W)
    p = "decoder.model.1.block.1"
    s, cin, cout, T = 3, 48, 24, 5
    x = rng.standard_normal((1, cin, T)).astype(np.float32)
    got = m.convt(p, torch.from_numpy(x), s).numpy()[0]
    v, g, b = W[p + ".weight_v"].astype(np.float64), W[p + ".weight_g"].astype(np.float64), W[p + ".bias"]
    w = g * v / (np.sqrt((v * v).sum(axis=(0, 1), keepdims=True)) + 1e-12)            # norm over all axes except the input one
    full = np.zeros((cout, (T - 1) * s + 2 * s))
    for n in range(T):
        for j in range(2 * s):
            full[:, n * s + j] += w[:, j, :] @ x[0, :, n]
    pad = 2                                                                             # ceil(3 / 2)
    ref = full[:, pad: full.shape[1] - (pad - 1)] + b[:, None]
    assert got.shape == ref.shape == (cout, od.convt_out_len(T, s)) and np.abs(got - ref).max() < 1e-5


def test_decode_shapes_and_range():
    cfg = od.TINY
    m = od.DacOracle(cfg, od.make_synthetic_weights(cfg))
    codes = np.random.default_rng(1).integers(0, cfg.codebook_size, (2, cfg.n_codebooks, 6))
    wav = m.decode_from_codes(codes)
    assert wav.shape == (2, od.num_samples(cfg, 6)) and np.abs(wav).max() <= 1.0 and wav.std() > 1e-3


def test_decoder_matches_hf_transformers_away_from_the_right_edge():
    """Independent implementation: HF transformers' DacModel decoder / quantizer with the oracle's (weight-norm folded) weights.
    HF's ConvTranspose1d has no output_padding, the reference passes output_padding = 1 (one more sample on the RIGHT of every
    block, pinned by the reference's own length tests), so the two agree sample by sample except near the right edge."""
    from transformers import DacConfig as HFC, DacModel
    cfg = od.DacConfig(encoder_dim=4, encoder_rates=(3, 5), latent_dim=24, decoder_dim=48, decoder_rates=(5, 3), n_codebooks=3,
                       codebook_size=32, codebook_dim=8)
    hc = HFC(encoder_hidden_size=4, downsampling_ratios=[3, 5], decoder_hidden_size=48, upsampling_ratios=[5, 3], n_codebooks=3,
             codebook_size=32, codebook_dim=8, hidden_size=24, sampling_rate=16000)
    hf = DacModel(hc).eval()
    W = od.make_synthetic_weights(cfg, seed=11)
    o = od.DacOracle(cfg, W)
    sd = hf.state_dict()

    def eff(p, transposed=False):                       # effective weight in torch layout
        w = od._wn(o.w[p + ".weight_g"], o.w[p + ".weight_v"], 2 if transposed else 0)
        return (w.permute(2, 0, 1) if transposed else w.permute(0, 2, 1)).contiguous()

    def put(hf_name, p, transposed=False):
        assert sd[hf_name + ".weight"].shape == eff(p, transposed).shape, (hf_name, p)
        sd[hf_name + ".weight"] = eff(p, transposed)
        sd[hf_name + ".bias"] = o.w[p + ".bias"]

    put("decoder.conv1", "decoder.model.0")
    for bi in range(2):
        p = f"decoder.model.{bi + 1}.block"
        sd[f"decoder.block.{bi}.snake1.alpha"] = o.w[p + ".0.alpha"].reshape(1, -1, 1)
        put(f"decoder.block.{bi}.conv_t1", p + ".1", transposed=True)
        for ri in range(3):
            q = f"{p}.{ri + 2}.block"
            h = f"decoder.block.{bi}.res_unit{ri + 1}"
            sd[h + ".snake1.alpha"] = o.w[q + ".0.alpha"].reshape(1, -1, 1)
            put(h + ".conv1", q + ".1")
            sd[h + ".snake2.alpha"] = o.w[q + ".2.alpha"].reshape(1, -1, 1)
            put(h + ".conv2", q + ".3")
    sd["decoder.snake1.alpha"] = o.w["decoder.model.3.alpha"].reshape(1, -1, 1)
    put("decoder.conv2", "decoder.model.4")
    for i in range(3):
        p = f"quantizer.quantizers.{i}"
        sd[f"{p}.codebook.weight"] = o.w[p + ".codebook.weight"]
        put(f"{p}.out_proj", p + ".outProj")
    hf.load_state_dict(sd)
    rng = np.random.default_rng(2)
    T = 40
    codes = rng.integers(0, 32, (2, 3, T))
    with torch.no_grad():
        z_hf = hf.quantizer.from_codes(torch.from_numpy(codes))[0]
        y_hf = hf.decoder(z_hf)[:, 0].numpy()
    z = o.from_codes(codes)
    assert torch.allclose(z, z_hf, atol=1e-5)
    y = o.decode_from_codes(codes)
    assert y.shape[1] == od.num_samples(cfg, T) and y_hf.shape[1] < y.shape[1]      # the reference keeps one more sample per block
    n = y_hf.shape[1] - 260                                                          # right-edge influence of the missing samples
    assert n > 300
    np.testing.assert_allclose(y[:, :n], y_hf[:, :n], rtol=1e-4, atol=2e-5)


def test_encoder_and_rvq_encode_match_hf_transformers():
    """Independent implementation for the encode side: HF DacModel.encoder / quantizer with the oracle's weights.  The encoder output
    agrees to float tolerance; codes agree wherever the oracle's top-2 distance margin is not a rounding-level tie."""
    from transformers import DacConfig as HFC, DacModel
    cfg = od.DacConfig(encoder_dim=4, encoder_rates=(3, 5), latent_dim=24, decoder_dim=48, decoder_rates=(5, 3), n_codebooks=3,
                       codebook_size=32, codebook_dim=8)
    hc = HFC(encoder_hidden_size=4, downsampling_ratios=[3, 5], decoder_hidden_size=48, upsampling_ratios=[5, 3], n_codebooks=3,
             codebook_size=32, codebook_dim=8, hidden_size=24, sampling_rate=16000)
    hf = DacModel(hc).eval()
    W = od.make_synthetic_weights(cfg, seed=5)
    o = od.DacOracle(cfg, W)
    sd = hf.state_dict()

    def put(hf_name, p):
        w = od._wn(o.w[p + ".weight_g"], o.w[p + ".weight_v"], 0).permute(0, 2, 1).contiguous()
        assert sd[hf_name + ".weight"].shape == w.shape, (hf_name, p)
        sd[hf_name + ".weight"] = w
        sd[hf_name + ".bias"] = o.w[p + ".bias"]

    put("encoder.conv1", "encoder.block.0")
    for bi in range(2):
        p = f"encoder.block.{bi + 1}.block"
        for ri in range(3):
            q, h = f"{p}.{ri}.block", f"encoder.block.{bi}.res_unit{ri + 1}"
            sd[h + ".snake1.alpha"] = o.w[q + ".0.alpha"].reshape(1, -1, 1)
            put(h + ".conv1", q + ".1")
            sd[h + ".snake2.alpha"] = o.w[q + ".2.alpha"].reshape(1, -1, 1)
            put(h + ".conv2", q + ".3")
        sd[f"encoder.block.{bi}.snake1.alpha"] = o.w[p + ".3.alpha"].reshape(1, -1, 1)
        put(f"encoder.block.{bi}.conv1", p + ".4")
    sd["encoder.snake1.alpha"] = o.w["encoder.block.3.alpha"].reshape(1, -1, 1)
    put("encoder.conv2", "encoder.block.4")
    for i in range(3):
        p = f"quantizer.quantizers.{i}"
        sd[f"{p}.codebook.weight"] = o.w[p + ".codebook.weight"]
        put(f"{p}.in_proj", p + ".inProj")
        put(f"{p}.out_proj", p + ".outProj")
    hf.load_state_dict(sd)
    rng = np.random.default_rng(3)
    audio = (0.3 * rng.standard_normal((2, 15 * 40))).astype(np.float32)
    codes, z, margins = o.encode(audio, return_latent=True, return_margins=True)
    with torch.no_grad():
        z_hf = hf.encoder(torch.from_numpy(audio)[:, None])
        codes_hf = hf.quantizer(z_hf)[1].numpy()
    assert z.shape == z_hf.shape == (2, 24, 40)
    np.testing.assert_allclose(z, z_hf.numpy(), rtol=1e-4, atol=1e-5)
    assert codes.shape == codes_hf.shape == (2, 3, 40)
    # a near-tie in an early codebook changes the residual of the later ones: compare up to the first near-tie per frame
    ok = np.cumprod(margins > 1e-4, axis=1).astype(bool)
    assert ok.mean() > 0.9 and (codes[ok] == codes_hf[ok]).all()


def test_encode_audio_pads_to_the_hop_and_round_trips_shapes():
    cfg = od.TINY
    o = od.DacOracle(cfg, od.make_synthetic_weights(cfg))
    audio = np.random.default_rng(4).standard_normal((2, 37)).astype(np.float32) * 0.2
    assert o.preprocess(audio).shape == (2, 1, 40)                                   # hop = 4
    codes = o.encode(audio)
    assert codes.shape == (2, cfg.n_codebooks, 10) and codes.min() >= 0 and codes.max() < cfg.codebook_size
    assert o.encode(audio, n_quantizers=2).shape == (2, 2, 10)
    np.testing.assert_array_equal(o.encode(audio, n_quantizers=2), codes[:, :2])
