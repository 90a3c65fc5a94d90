"""Orpheus <-> SNAC token framing (integer-exact work).  Test infrastructure only.

Restates, in numpy / plain Python:
  * OrpheusTokens                      LlamaTTS.swift:20-30
  * llamaDecodeAudioFromCodes (split)  LlamaTTS.swift:41-64   -> deinterleave()
  * llamaEncodeAudioToCodes (merge)    LlamaTTS.swift:84-97   -> interleave()
  * LlamaTTSModel.parseOutput          LlamaTTS.swift:383-434 -> parse_output()
  * prompt wrapping of prepareInputIds LlamaTTS.swift:477-552 -> wrap_prompt()
(paths relative to Sources/MLXAudioTTS/Models/Llama/).
"""
from __future__ import annotations

import numpy as np

START_OF_HUMAN = 128259
END_OF_HUMAN = 128260
END_OF_TEXT = 128009
START_OF_SPEECH = 128257
END_OF_SPEECH = 128258
PAD_TOKEN = 128263
AUDIO_START = 128261
AUDIO_END = 128262
AUDIO_TOKEN_OFFSET = 128266
CODEBOOK = 4096


def deinterleave(code_list):
    """LlamaTTS.swift:41-58.  7-token frames -> (L0[G], L1[2G], L2[4G]).

    numGroups = (count + 1) / 7 exactly as the reference (:46); callers hand in
    lists already trimmed to a multiple of 7 (parseOutput :424)."""
    code_list = [int(c) for c in code_list]
    n_groups = (len(code_list) + 1) // 7
    l1, l2, l3 = [], [], []
    for i in range(n_groups):
        b = 7 * i
        l1.append(code_list[b])
        l2.append(code_list[b + 1] - 4096)
        l3.append(code_list[b + 2] - 2 * 4096)
        l3.append(code_list[b + 3] - 3 * 4096)
        l2.append(code_list[b + 4] - 4 * 4096)
        l3.append(code_list[b + 5] - 5 * 4096)
        l3.append(code_list[b + 6] - 6 * 4096)
    return (np.asarray(l1, np.int32), np.asarray(l2, np.int32), np.asarray(l3, np.int32))


def interleave(l1, l2, l3):
    """LlamaTTS.swift:84-95 (exact inverse of deinterleave)."""
    out = []
    for i in range(len(l1)):
        out += [int(l1[i]), int(l2[2 * i]) + 4096, int(l3[4 * i]) + 2 * 4096,
                int(l3[4 * i + 1]) + 3 * 4096, int(l2[2 * i + 1]) + 4 * 4096,
                int(l3[4 * i + 2]) + 5 * 4096, int(l3[4 * i + 3]) + 6 * 4096]
    return np.asarray(out, np.int32)


def parse_output_row(ids):
    """parseOutput for ONE utterance (LlamaTTS.swift:383-434 with B = 1, which is the
    only way the reference ever calls it, :749-752): crop after the LAST start-of-speech,
    drop end-of-speech, trim to a multiple of 7, subtract the audio token offset."""
    ids = [int(t) for t in ids]
    last = None
    for j, t in enumerate(ids):
        if t == START_OF_SPEECH:
            last = j
    cropped = ids[last + 1:] if last is not None else ids
    kept = [t for t in cropped if t != END_OF_SPEECH]
    n = (len(kept) // 7) * 7
    return np.asarray([t - AUDIO_TOKEN_OFFSET for t in kept[:n]], np.int32)


def wrap_prompt(text_ids):
    """[SOH] text [EOT][EOH]   (LlamaTTS.swift:478-482,533-537)."""
    return np.asarray([START_OF_HUMAN] + [int(t) for t in text_ids] + [END_OF_TEXT, END_OF_HUMAN], np.int32)


def left_pad_batch(rows):
    """Left-pad with 128263 to the longest row; mask = ids != pad (LlamaTTS.swift:493-552)."""
    m = max(len(r) for r in rows)
    ids = np.full((len(rows), m), PAD_TOKEN, np.int32)
    for i, r in enumerate(rows):
        ids[i, m - len(r):] = r
    return ids, ids != PAD_TOKEN


# ---- VyvoTTS (Qwen3 LM + SNAC), Sources/MLXAudioTTS/Models/Qwen3/Qwen3.swift:19-29
VYVO_START_OF_SPEECH, VYVO_END_OF_SPEECH, VYVO_START_OF_AI, VYVO_AUDIO_OFFSET = 151670, 151671, 151674, 151679


def parse_output_row_vyvo(tokens):
    """Qwen3Model.parseOutputRow (Qwen3.swift:332-358): crop after the last START_OF_SPEECH, else after the token before the
    first audio token that follows the last START_OF_AI, else nothing cropped; drop END_OF_SPEECH; trim to 7k; subtract."""
    t = [int(x) for x in tokens]
    start = None
    if VYVO_START_OF_SPEECH in t:
        start = len(t) - 1 - t[::-1].index(VYVO_START_OF_SPEECH)
    elif VYVO_START_OF_AI in t:
        soa = len(t) - 1 - t[::-1].index(VYVO_START_OF_AI)
        for j in range(soa + 1, len(t)):
            if t[j] >= VYVO_AUDIO_OFFSET:
                start = j - 1
                break
    sl = t[start + 1:] if start is not None else t
    f = [x for x in sl if x != VYVO_END_OF_SPEECH]
    n = (len(f) // 7) * 7
    return np.asarray([x - VYVO_AUDIO_OFFSET for x in f[:n]], np.int32)


def decode_audio_from_codes_chunked(code_list, snac_oracle, chunk_groups=50, noises=None):
    """VyvoTTS decodeAudioFromCodes (Qwen3.swift:47-83): at most `chunk_groups` groups -> one SNAC decode; otherwise the groups are
    decoded chunk by chunk, every chunk an INDEPENDENT decode (decodeAudioChunk :85-114 = de-interleave + snacModel.decode), and
    the samples are concatenated.  `noises` (optional, one [1, T_i] array per decoder block, sized for the whole utterance) is
    sliced at the chunk's offset so that a test can inject the same noise on both sides."""
    code_list = [int(c) for c in code_list]
    n_groups = (len(code_list) + 1) // 7                                   # :48

    def one(codes, nz):
        l0, l1, l2 = deinterleave(codes)
        return snac_oracle.decode([l0[None], l1[None], l2[None]], nz)[0, 0]

    if n_groups <= chunk_groups:                                           # :51-55
        return one(code_list, noises)
    out = []
    start = 0
    while start < n_groups:                                                # :63-80
        end = min(start + chunk_groups, n_groups)
        nz = None
        if noises is not None:
            nz = []
            for z in noises:
                per_group = z.shape[-1] // n_groups
                nz.append(z[..., start * per_group: end * per_group])
        out.append(one(code_list[7 * start: min(7 * end, len(code_list))], nz))
        start = end
    return np.concatenate(out)
