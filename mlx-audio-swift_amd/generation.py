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
    frame_constrained: int | bool = False  # synthetic-weight benches only: 1 / True narrow sampler, 2 full-vocabulary sampler (mi_speech.h)
    row_offset: int = 0
    sampler_flavor: int = 0          # 0 mlx-lm sampler + RepetitionContext; 1 Soprano (Soprano.swift:888-901)

    def to_c(self) -> "_lib.GenParamsC":
        return _lib.GenParamsC(int(self.max_tokens), float(self.temperature), float(self.top_p),
                               float(self.repetition_penalty or 0.0), int(self.repetition_context_size), int(self.seed),
                               int(self.frame_constrained), int(self.row_offset), int(self.sampler_flavor), 0)


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


def stream_events(start_call, decode_event, cancel_flag=None):
    """Run a blocking `mis_*_generate_stream` C call on a worker thread and yield its events as the callback fires
    (the Swift shim does the same hop into an AsyncThrowingStream continuation, LlamaTTS.swift:792-911).
    start_call(cb, cancel_flag_address) -> status runs the C call; decode_event(row, kind, payload, n) -> event object.
    Closing the generator early sets the cancel flag (continuation.onTermination -> task.cancel()); `cancel_flag` (a
    ctypes.c_int the caller may set from another thread) is used instead of a private flag when given."""
    import ctypes as C
    import queue
    import threading

    q: "queue.Queue" = queue.Queue()
    flag = cancel_flag if cancel_flag is not None else C.c_int(0)
    DONE = object()

    def cb(user, row, kind, payload, n):
        q.put(decode_event(row, kind, payload, n))

    cbf = _lib.EVENT_CB(cb)

    def run():
        try:
            st = start_call(cbf, C.addressof(flag))
            q.put((DONE, st, _lib.last_error() if st != _lib.MIS_OK else ""))     # last_error is thread-local: read it here
        except BaseException as e:  # pragma: no cover
            q.put((DONE, -1, repr(e)))

    th = threading.Thread(target=run, daemon=True)
    th.start()
    finished = False
    try:
        while True:
            ev = q.get()
            if isinstance(ev, tuple) and len(ev) == 3 and ev[0] is DONE:
                finished = True                       # the C call has returned: nothing left to cancel
                if ev[1] != _lib.MIS_OK:
                    raise AudioGenerationError(ev[1] if ev[1] >= 0 else 2, ev[2])
                return
            yield ev
    finally:
        if finished:
            th.join()
        else:
            # early close (GeneratorExit / an exception in the consumer): stop the worker.  A caller-supplied flag is only borrowed for
            # that - its value is put back once the worker has returned, so a flag reused across calls does not cancel the next one
            prev = flag.value
            flag.value = 1
            th.join()
            if cancel_flag is not None:
                flag.value = prev


def decode_audio_event(row, kind, payload, n):
    """payload of the three AudioGeneration cases (GenerationTypes.swift:50-61) -> host events"""
    import ctypes as C
    if kind == _lib.EVENT_TOKEN:
        return TokenEvent(row, C.cast(payload, C.POINTER(C.c_int32))[0])
    if kind == _lib.EVENT_INFO:
        i = C.cast(payload, C.POINTER(_lib.GenInfoC))[0]
        return InfoEvent(row, AudioGenerationInfo(i.prompt_token_count, i.generation_token_count, i.prefill_time,
                                                  i.generate_time, i.tokens_per_second, i.peak_memory_gb))
    a = np.ctypeslib.as_array(C.cast(payload, C.POINTER(C.c_float)), shape=(max(n, 1),))[:n].copy()
    return AudioEvent(row, a)
