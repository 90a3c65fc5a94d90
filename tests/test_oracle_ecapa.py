"""CPU: pins for oracle/ecapa.py (the Qwen3-TTS speaker encoder, not yet built on the device): an independent torch implementation
(nn.Conv1d with torch's own reflect padding, torch.var_mean / softmax) and structural properties."""
import numpy as np
import torch
import torch.nn.functional as TF

from oracle import ecapa as oe


def _torch_ecapa(cfg, W, mel):
    w = {k: torch.from_numpy(v) for k, v in W.items()}

    def tdnn(p, x, k, d):
        pad = (k - 1) * d // 2
        if pad:
            x = TF.pad(x, (pad, pad), mode="reflect")
        return TF.relu(TF.conv1d(x, w[p + ".conv.weight"].permute(0, 2, 1), w[p + ".conv.bias"], dilation=d))

    def block(p, x, k, d):
        y = tdnn(p + ".tdnn1", x, 1, 1)
        parts, prev = [], None
        for i, c in enumerate(torch.chunk(y, cfg.enc_res2net_scale, dim=1)):
            prev = c if i == 0 else tdnn(f"{p}.res2net_block.blocks.{i - 1}", c if i == 1 else c + prev, k, d)
            parts.append(prev)
        y = tdnn(p + ".tdnn2", torch.cat(parts, 1), 1, 1)
        g = TF.conv1d(TF.relu(TF.conv1d(y.mean(2, keepdim=True), w[p + ".se_block.conv1.weight"].permute(0, 2, 1), w[p + ".se_block.conv1.bias"])),
                      w[p + ".se_block.conv2.weight"].permute(0, 2, 1), w[p + ".se_block.conv2.bias"])
        return y * torch.sigmoid(g) + x
    x = torch.from_numpy(mel).transpose(1, 2)
    x = tdnn("blocks.0", x, cfg.enc_kernel_sizes[0], cfg.enc_dilations[0])
    hs = []
    for i in range(1, len(cfg.enc_channels) - 1):
        x = block(f"blocks.{i}", x, cfg.enc_kernel_sizes[i], cfg.enc_dilations[i])
        hs.append(x)
    x = tdnn("mfa", torch.cat(hs, 1), cfg.enc_kernel_sizes[-1], cfg.enc_dilations[-1])
    var, mu = torch.var_mean(x, dim=2, keepdim=True, unbiased=False)
    a = torch.cat([x, mu.expand_as(x), torch.sqrt(var + 1e-12).expand_as(x)], 1)
    a = torch.softmax(TF.conv1d(torch.tanh(tdnn("asp.tdnn", a, 1, 1)), w["asp.conv.weight"].permute(0, 2, 1), w["asp.conv.bias"]), dim=2)
    m = (a * x).sum(2, keepdim=True)
    sd = torch.sqrt(torch.clamp((a * (x - m) ** 2).sum(2, keepdim=True), min=1e-12))
    return TF.conv1d(torch.cat([m, sd], 1), w["fc.weight"].permute(0, 2, 1), w["fc.bias"])[:, :, 0].numpy()


def test_matches_independent_torch_implementation():
    for cfg, T in ((oe.TINY, 37), (oe.TINY, 9), (oe.EcapaConfig(mel_dim=16, enc_dim=32, enc_channels=(32, 32, 32, 32, 96), enc_attention_channels=16,
                                                                 enc_res2net_scale=8, enc_se_channels=8), 50)):
        W = oe.make_synthetic_weights(cfg)
        mel = np.random.default_rng(T).standard_normal((2, T, cfg.mel_dim)).astype(np.float32)
        got = oe.EcapaOracle(cfg, W)(mel)
        ref = _torch_ecapa(cfg, W, mel)
        assert got.shape == ref.shape == (2, cfg.enc_dim)
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5)


def test_reflect_pad_rule_and_row_independence():
    x = np.arange(10, dtype=np.float32).reshape(1, 1, 10)
    assert oe.reflect_pad(x, 2)[0, 0].tolist() == [2, 1] + list(range(10)) + [8, 7]
    assert oe.reflect_pad(x[..., :2], 4)[0, 0].tolist() == [1, 0, 1, 0]          # pad clamped to T - 1
    assert oe.reflect_pad(x[..., :1], 3).shape == (1, 1, 1)                         # T <= 1: unchanged
    cfg = oe.TINY
    o = oe.EcapaOracle(cfg, oe.make_synthetic_weights(cfg))
    mel = np.random.default_rng(0).standard_normal((3, 20, cfg.mel_dim)).astype(np.float32)
    full = o(mel)
    assert np.allclose(o(mel[1:2]), full[1:2], atol=1e-6)                           # rows are independent
