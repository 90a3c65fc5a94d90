"""-m gpu: EnCodec decoder ((f)1) against oracle/encodec.py: conv stem, persistent LSTM, upsampling blocks, waveform, chunked decode."""
import numpy as np
import pytest

import mlx_audio_swift_amd as mas
from oracle import encodec as oe

pytestmark = pytest.mark.gpu


def _pair(ocfg, **extra):
    W = oe.make_synthetic_weights(ocfg)
    fields = {k: getattr(ocfg, k) for k in mas.EncodecConfig.__dataclass_fields__ if hasattr(ocfg, k)}
    fields.update(extra)
    return oe.EncodecOracle(ocfg, W), mas.Encodec.from_weights(mas.EncodecConfig(**fields), W)


# the second config has an LSTM wide enough to be split over several blocks (grid barrier per step) and 2 residual layers? no:
# dilation > 1 is not built; it exercises hidden 128 -> 8 blocks of 16 units
WIDE = oe.EncodecConfig(num_filters=32, codebook_size=64, codebook_dim=32, hidden_size=32, upsampling_ratios=(2, 2),
                        target_bandwidths=(8.0,), sampling_rate=1600)


# the 48 kHz family at a width where the transposed convs and the k = 7 convs take the split-bf16 path as well
WIDE_48K = oe.EncodecConfig(audio_channels=2, num_filters=32, codebook_size=64, codebook_dim=32, hidden_size=32, upsampling_ratios=(4, 2),
                            target_bandwidths=(8.0,), sampling_rate=1600, use_causal_conv=False, norm_type="time_group_norm")


@pytest.mark.parametrize("ocfg", [oe.TINY, WIDE, oe.TINY_48K, WIDE_48K],
                         ids=["tiny-1block-lstm", "wide-multiblock-lstm", "48k-family-tiny", "48k-family-wide"])
def test_decode_frame_matches_oracle(ocfg):
    orc, dev = _pair(ocfg)
    rng = np.random.default_rng(0)
    nq = ocfg.num_quantizers
    for B, T in ((2, 9), (1, 1), (1, 40)):
        codes = rng.integers(0, ocfg.codebook_size, (B, nq, T)).astype(np.int32)
        stages = [(1, "conv0"), (2, "lstm")] + [(3 + i, f"block{i}") for i in range(len(ocfg.upsampling_ratios))]
        for sid, name in stages:
            ref = orc.decode_frame(codes, stop_after=name)
            got = dev.debug_tap(codes, sid)
            assert got.shape == ref.shape, (name, got.shape, ref.shape)
            assert np.abs(got - ref).max() <= 3e-4 * max(np.abs(ref).max(), 1e-3), (name, B, T)
        ref = orc.decode_frame(codes, scale=[0.5] * B)
        got = dev.decode_frame(codes, scale=0.5)
        want = (B, T * ocfg.hop_length) if ocfg.audio_channels == 1 else (B, ocfg.audio_channels, T * ocfg.hop_length)
        assert got.shape == ref.shape == want and np.abs(got - ref).max() <= 3e-4 * max(np.abs(ref).max(), 1e-3)
    # fewer quantizers than the model holds (bandwidth selection, EncodecQuantization.swift:66-74)
    codes = rng.integers(0, ocfg.codebook_size, (1, 1, 5)).astype(np.int32)
    assert np.abs(dev.decode_frame(codes) - orc.decode_frame(codes)).max() < 1e-3


def test_chunked_decode_overlap_add():
    ocfg = oe.TINY
    orc, dev = _pair(ocfg, chunk_length_s=0.1, overlap=0.25)              # 120-sample chunks, stride 90
    rng = np.random.default_rng(1)
    chunks = rng.integers(0, ocfg.codebook_size, (3, 2, ocfg.num_quantizers, 10)).astype(np.int32)
    ref = oe.linear_overlap_add([orc.decode_frame(chunks[i]) for i in range(3)], 90)
    got = dev.decode(chunks)
    assert got.shape == ref.shape == (2, 90 * 2 + 120) and np.abs(got - ref).max() <= 3e-4 * np.abs(ref).max()
    one = mas.Encodec.from_weights(mas.EncodecConfig(**{k: getattr(ocfg, k) for k in mas.EncodecConfig.__dataclass_fields__ if hasattr(ocfg, k)}),
                                   oe.make_synthetic_weights(ocfg))
    with pytest.raises(mas.AudioGenerationError):
        one.decode(chunks)                                                  # "Expected one frame" without chunking (:373-376)


def test_chunked_stereo_decode_of_the_48k_family():
    ocfg = oe.TINY_48K
    orc, dev = _pair(ocfg, chunk_length_s=0.1, overlap=0.25)              # 120-sample chunks, stride 90
    chunks = np.random.default_rng(2).integers(0, ocfg.codebook_size, (3, 2, ocfg.num_quantizers, 10)).astype(np.int32)
    scales = [np.array([0.5, 2.0], np.float32)] * 3
    frames = [orc.decode_frame(chunks[i], scale=scales[i]) for i in range(3)]                      # [B, 2, 120]
    ref = np.stack([oe.linear_overlap_add([f[:, ch] for f in frames], 90) for ch in range(2)], axis=1)
    got = dev.decode(chunks, audio_scales=scales)
    assert got.shape == ref.shape == (2, 2, 90 * 2 + 120) and np.abs(got - ref).max() <= 3e-4 * np.abs(ref).max()
