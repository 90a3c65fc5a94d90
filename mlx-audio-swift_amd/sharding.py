"""Utterance-batch data parallelism: rows are independent units, weights are replicated,
each rank (one process per GPU) generates a contiguous block of rows, and the only exchange is ONE
all-gather of the decoded PCM (+ lengths) at the end - RCCL over xGMI when the backend is "nccl",
gloo in the CPU tests.  The reference has no counterpart (single device)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_rows(n_rows: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one row."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_pcm(pcm: torch.Tensor, lens: torch.Tensor, n_rows: int):
    """pcm [rows_local, stride] float32, lens [rows_local] int64 (same device).  Returns
    (pcm_all [n_rows, stride], lens_all [n_rows]) on every rank.  Ranks may own different row counts:
    blocks are padded to the largest before the fixed-size all-gather."""
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
