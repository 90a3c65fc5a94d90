"""Utterance-batch data parallelism: rows are independent units, weights are replicated, each shard generates a
contiguous block of rows with the RNG keyed by the global row, and the only exchange is ONE all-gather of the decoded PCM
(+ lengths) at the end.  The reference has no counterpart (single device).

The data path lives BEHIND the C ABI (csrc/group.hip), so a Swift host has it too:
  * TTSGroup      one process, N replicas (one LlamaTTSModel + SNAC per device): mis_tts_group_generate[_device] - a worker thread
                  and stream per GPU inside the library, all-gather by direct peer copies over xGMI;
  * Communicator  one process per GPU (torchrun & co): mis_comm_* - an RCCL communicator created from a 128-byte id the host
                  broadcasts by any means, ncclAllGather of the PCM blocks on the library's stream.
all_gather_pcm() below keeps the torch.distributed form for hosts that hold their PCM in torch tensors (and is what the
world-2 gloo CPU test runs)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .generation import GenerateParameters, check


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one row (== mis_shard_rows)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class TTSGroup:
    """mis_group: `replicas` = finalized LlamaTTSModel objects (each with its own codec), one per device (or several logical
    shards of one device)."""

    def __init__(self, replicas):
        self.replicas = list(replicas)
        arr = (C.c_void_p * len(self.replicas))(*[r._h for r in self.replicas])
        h = C.c_void_p()
        check(_lib.lib().mis_tts_group_create(arr, len(self.replicas), C.byref(h)))
        self._h = h

    def __len__(self):
        return _lib.lib().mis_tts_group_size(self._h)

    def generate_batch(self, prompt_rows, generation_parameters: GenerateParameters | None = None, return_tokens: bool = False):
        """LlamaTTSModel.generate_batch over the group: same outputs, rows sharded inside the library."""
        from .tts import LlamaTTSModel
        gp = generation_parameters or GenerateParameters()
        flat, lens = LlamaTTSModel._flatten(prompt_rows)
        B = len(lens)
        gpc = gp.to_c()
        pcm = C.c_void_p(); stride = C.c_int64(); plens = (C.c_int64 * B)()
        toks = C.c_void_p(); tstride = C.c_int64(); ntok = (C.c_int32 * B)()
        check(_lib.lib().mis_tts_group_generate(self._h, flat.ctypes.data, lens.ctypes.data, B, C.byref(gpc), C.byref(pcm), C.byref(stride),
                                                plens, C.byref(toks) if return_tokens else None, C.byref(tstride), ntok))
        try:
            arr = np.ctypeslib.as_array(C.cast(pcm, C.POINTER(C.c_float)), shape=(B, max(stride.value, 1)))
            out = [arr[b, : plens[b]].copy() for b in range(B)]
            if return_tokens:
                t = np.ctypeslib.as_array(C.cast(toks, C.POINTER(C.c_int32)), shape=(B, max(tstride.value, 1)))
                tok = [t[b, : ntok[b]].copy() for b in range(B)]
        finally:
            _lib.lib().mis_free(pcm)
            if return_tokens and toks:
                _lib.lib().mis_free(toks)
        return (out, tok) if return_tokens else out

    def generate_device(self, prompt_rows, generation_parameters, pcm_ptrs, pcm_stride: int):
        """PCM left in HBM and all-gathered: pcm_ptrs[i] = device pointer of replica i's [batch, pcm_stride] float32 buffer.
        Returns (pcm_lens, n_tokens)."""
        from .tts import LlamaTTSModel
        flat, lens = LlamaTTSModel._flatten(prompt_rows)
        B = len(lens)
        gpc = generation_parameters.to_c()
        ptrs = (C.c_void_p * len(self.replicas))(*[int(p) for p in pcm_ptrs])
        plens = (C.c_int64 * B)(); ntok = (C.c_int32 * B)()
        check(_lib.lib().mis_tts_group_generate_device(self._h, flat.ctypes.data, lens.ctypes.data, B, C.byref(gpc), ptrs, int(pcm_stride),
                                                       plens, ntok))
        return list(plens), list(ntok)

    def last_timing(self) -> dict:
        t = _lib.GroupTimingC()
        check(_lib.lib().mis_tts_group_last_timing(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in t._fields_}

    def close(self):
        if self._h is not None:
            _lib.lib().mis_tts_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Communicator:
    """mis_comm: RCCL communicator of one-process-per-GPU jobs.  `broadcast` ships rank 0's 128-byte id to the other ranks
    (bootstrap only - e.g. torch.distributed.broadcast_object_list, MPI, a shared file); the PCM itself only ever moves through
    mis_comm_all_gather_pcm (ncclAllGather over xGMI)."""
    ID_BYTES = 128

    def __init__(self, device: int, rank: int, world: int, broadcast=None):
        self.rank, self.world, self.device = rank, world, device
        buf = C.create_string_buffer(self.ID_BYTES)
        if rank == 0:
            check(_lib.lib().mis_comm_unique_id(buf))
        raw = bytes(buf.raw)
        if world > 1:
            if broadcast is None:
                raise ValueError("a broadcast(bytes, root=0) -> bytes function is needed when world > 1")
            raw = broadcast(raw)
        h = C.c_void_p()
        check(_lib.lib().mis_comm_create(device, rank, world, raw, C.byref(h)))
        self._h = h

    def all_gather_pcm(self, pcm_local_ptr: int, lens_local, rows_local: int, stride: int, pcm_all_ptr: int):
        """Device pointers in / out; returns (lens_all [world * rows_local], device milliseconds of the exchange)."""
        ll = np.ascontiguousarray(lens_local, dtype=np.int64)
        out = np.zeros(self.world * rows_local, np.int64)
        ms = C.c_double()
        check(_lib.lib().mis_comm_all_gather_pcm(self._h, int(pcm_local_ptr), ll.ctypes.data, rows_local, int(stride), int(pcm_all_ptr),
                                                 out.ctypes.data, C.byref(ms)))
        return out, ms.value

    def close(self):
        if self._h is not None:
            _lib.lib().mis_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def all_gather_pcm(pcm, lens, n_rows: int):
    """torch.distributed form (hosts that keep PCM in torch tensors; gloo on CPU): pcm [rows_local, stride] float32, lens
    [rows_local] int64 (same device).  Returns (pcm_all [n_rows, stride], lens_all [n_rows]) on every rank.  Ranks may own
    different row counts: blocks are padded to the largest before the fixed-size all-gather."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return pcm, lens
    world = dist.get_world_size()
    per = max(shard_rows(n_rows, r, world)[1] - shard_rows(n_rows, r, world)[0] for r in range(world))
    stride = pcm.shape[1]
    buf = torch.zeros((per, stride), dtype=pcm.dtype, device=pcm.device)
    buf[: pcm.shape[0]] = pcm
    lbuf = torch.zeros((per,), dtype=lens.dtype, device=lens.device)
    lbuf[: lens.shape[0]] = lens
    out = torch.empty((world * per, stride), dtype=pcm.dtype, device=pcm.device)
    lout = torch.empty((world * per,), dtype=lens.dtype, device=lens.device)
    dist.all_gather_into_tensor(out, buf)
    dist.all_gather_into_tensor(lout, lbuf)
    rows, ls = [], []
    for r in range(world):
        lo, hi = shard_rows(n_rows, r, world)
        rows.append(out[r * per: r * per + (hi - lo)])
        ls.append(lout[r * per: r * per + (hi - lo)])
    return torch.cat(rows, 0), torch.cat(ls, 0)
