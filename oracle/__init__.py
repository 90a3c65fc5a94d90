"""CPU oracle for the mlx-audio-swift TTS/codec hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product path (``mlx-audio-swift_amd`` -> ``libmi_speech.so``) never calls into
this package and fails loudly when the HIP library is missing.

PARITY UNPINNED: the reference (Swift + MLX/Metal, deps un-vendored:
mlx-swift 0.31.4, mlx-swift-lm 3.31.4, Package.swift:63-66) cannot be built,
imported or run in this environment, and its own tests pin no numeric result
on the Orpheus / SNAC path (Tests/MLXAudioSmokeTests.swift:78-110,246-335 only
assert "non-empty").  Every function here is therefore a *restatement* of the
reference's Swift source, citing file:line, cross-checked against independent
implementations (torch.nn.functional convs, HF transformers Llama /
WhisperFeatureExtractor) and the few known-answer facts the reference does
hold (integer (de)interleave inverse pair, token constants, Hamming spot
values, all-zero log-mel = -1.5).
"""
