"""Generation types shared by the host mirror.  Mirrors
Sources/MLXAudioCore/Generation/GenerationTypes.swift:14-128 and the GenerateParameters fields the
Orpheus loop reads (LlamaTTS.swift:573-581)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _lib


class AudioGenerationError(Exception):
    """AudioGenerationError (GenerationTypes.swift:66-87); `.case` holds the Swift case name."""

    def __init__(self, status: int, message: str):
        self.status = status
        self.case = _lib.STATUS_NAMES.get(status, str(status))
        super().__init__(f"{self.case}: {message}")


def check(status: int):
    if status != _lib.MIS_OK:
        raise AudioGenerationError(status, _lib.last_error())


@dataclass
class GenerateParameters:
    """defaultGenerationParameters, LlamaTTS.swift:573-581"""
    max_tokens: int = 1200
    temperature: float = 0.6
    top_p: float = 0.8
    repetition_penalty: float = 1.3
    repetition_context_size: int = 20
    seed: int = 0                    # engine RNG key (MLX uses its global stream)
    frame_constrained: bool = False  # synthetic-weight benches only (see include/mi_speech.h)
    row_offset: int = 0
    sampler_flavor: int = 0          # 0 mlx-lm sampler + RepetitionContext; 1 Soprano (Soprano.swift:888-901)

    def to_c(self) -> "_lib.GenParamsC":
        return _lib.GenParamsC(int(self.max_tokens), float(self.temperature), float(self.top_p),
                               float(self.repetition_penalty or 0.0), int(self.repetition_context_size), int(self.seed),
                               1 if self.frame_constrained else 0, int(self.row_offset), int(self.sampler_flavor), 0)


@dataclass
class AudioGenerationInfo:
    """GenerationTypes.swift:14-45"""
    prompt_token_count: int
    generation_token_count: int
    prefill_time: float
    generate_time: float
    tokens_per_second: float
    peak_memory_usage: float


# AudioGeneration enum cases (GenerationTypes.swift:50-61)
@dataclass
class TokenEvent:
    row: int
    token: int


@dataclass
class InfoEvent:
    row: int
    info: AudioGenerationInfo


@dataclass
class AudioEvent:
    row: int
    audio: np.ndarray
