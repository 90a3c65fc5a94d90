"""world_size-2 gloo test of the multi-GPU path: row sharding + the single PCM all-gather."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mlx_audio_swift_amd.sharding import all_gather_pcm, shard_rows


def test_shard_rows_partition():
    for n in (1, 7, 32, 33):
        for w in (1, 2, 3, 8):
            blocks = [shard_rows(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def _fake_generate(row: int, stride: int):
    n = 10 + 3 * row
    x = torch.zeros(stride)
    x[:n] = torch.arange(n, dtype=torch.float32) + 1000 * row
    return x, n


def _worker(rank, world, n_rows, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_rows(n_rows, rank, world)
    stride = 64
    rows = [_fake_generate(r, stride) for r in range(lo, hi)]
    pcm = torch.stack([r[0] for r in rows]) if rows else torch.zeros((0, stride))
    lens = torch.tensor([r[1] for r in rows], dtype=torch.int64)
    allp, alll = all_gather_pcm(pcm, lens, n_rows)
    q.put((rank, allp.numpy(), alll.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_pcm_world2_gloo():
    n_rows, world, port = 5, 2, 29517
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, n_rows, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = torch.stack([_fake_generate(r, 64)[0] for r in range(n_rows)]).numpy()
    for rank, allp, alll in res:
        assert allp.shape == (n_rows, 64)
        assert np.array_equal(allp, ref)
        assert alll.tolist() == [10 + 3 * r for r in range(n_rows)]


def test_c_abi_shard_rows_equals_host_mirror_and_comm_fails_loudly_without_gpu():
    import ctypes as C
    import pytest
    import mlx_audio_swift_amd as mas
    from mlx_audio_swift_amd import _lib
    L = _lib.lib()
    for n in (0, 1, 7, 32, 33, 256):
        for w in (1, 2, 3, 8):
            for r in range(w):
                lo, hi = C.c_int(-1), C.c_int(-1)
                L.mis_shard_rows(n, r, w, C.byref(lo), C.byref(hi))
                assert (lo.value, hi.value) == shard_rows(n, r, w)
    if L.mis_device_count() == 0:                      # no GPU here: the communicator must refuse, not fall back
        from mlx_audio_swift_amd.sharding import Communicator
        with pytest.raises(mas.AudioGenerationError) as e:
            Communicator(0, 0, 1)
        assert e.value.case == "device"
