"""world_size-2 gloo test (CPU) of the N>1 path: contiguous batch partition, scatter from rank 0,
per-rank work on the shard, gather back.  The per-rank "work" here is a pure-Python stand-in that
tags every polynomial with its global index (the HIP transforms need a GPU and are covered by the
-m gpu tests); what is checked is that every polynomial is processed exactly once, in place, by the
rank that owns it, with no data-path collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from concrete_ntt_amd import shard


def test_shard_bounds_cover_the_batch_exactly():
    for batch in (0, 1, 2, 7, 64, 65536, 1 << 20):
        for world in (1, 2, 3, 4, 8):
            spans = [shard.shard_bounds(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
                assert e0 == b1 and e0 >= b0
            sizes = shard.shard_sizes(batch, world)
            assert sum(sizes) == batch and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.shard_bounds(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, batch, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(batch * n, dtype=torch.int64) if rank == 0 else torch.empty(0, dtype=torch.int64)
        mine = shard.scatter_batch(full, n, src=0)
        b, e = shard.shard_bounds(batch, world, rank)
        assert mine.numel() == (e - b) * n
        assert torch.equal(mine, torch.arange(b * n, e * n, dtype=torch.int64))
        # "transform" the shard in place: polynomial k (global index) -> value + 1000 * (k + 1)
        view = mine.view(e - b, n)
        view += 1000 * (torch.arange(b, e, dtype=torch.int64) + 1).unsqueeze(1)
        got = shard.gather_batch(mine, n, batch, dst=0)
        if rank == 0:
            want = torch.arange(batch * n, dtype=torch.int64).view(batch, n)
            want = want + 1000 * (torch.arange(batch, dtype=torch.int64) + 1).unsqueeze(1)
            out.put(bool(torch.equal(got.view(batch, n), want)))
        else:
            assert got is None
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [5, 8])
def test_scatter_work_gather_world2(batch):
    world, n = 2, 16
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, n, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get() is True
