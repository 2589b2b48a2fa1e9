"""Batch sharding across the GPUs of one node.

Polynomials are independent units, so a batch of B polynomials is split into contiguous shards
(rank r owns [r*B/W, (r+1)*B/W), remainders to the low ranks) and every rank transforms its own shard
with its own plan replica.  There is NO collective on the data path.  torch.distributed (backend
"nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests) is used only for the optional
scatter / gather when the batch starts or ends on one rank.
"""


def shard_bounds(batch, world, rank):
    """[begin, end) polynomial indices of `rank`'s contiguous shard."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(batch, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_sizes(batch, world):
    return [shard_bounds(batch, world, r)[1] - shard_bounds(batch, world, r)[0] for r in range(world)]


def scatter_batch(full, n, src=0, group=None):
    """Distribute `full` (B*n elements on rank `src`, ignored elsewhere) into per-rank shards.
    Returns this rank's shard tensor.  Point-to-point sends from the root: on xGMI each of the 7 links
    carries one shard concurrently."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    meta = torch.zeros(1, dtype=torch.int64, device=full.device if rank == src else _dev(full))
    if rank == src:
        meta[0] = full.numel() // n
    dist.broadcast(meta, src=src, group=group)
    batch = int(meta.item())
    b, e = shard_bounds(batch, world, rank)
    if rank == src:
        mine = full[b * n:e * n].clone()
        reqs = []
        for r in range(world):
            if r == src:
                continue
            rb, re_ = shard_bounds(batch, world, r)
            if re_ > rb:
                reqs.append(dist.isend(full[rb * n:re_ * n].contiguous(), dst=r, group=group))
        for q in reqs:
            q.wait()
        return mine
    mine = torch.empty((e - b) * n, dtype=full.dtype, device=full.device)
    if e > b:
        dist.recv(mine, src=src, group=group)
    return mine


def gather_batch(shard, n, batch, dst=0, group=None):
    """Inverse of scatter_batch: returns the full batch on rank `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if rank != dst:
        if shard.numel():
            dist.send(shard.contiguous(), dst=dst, group=group)
        return None
    full = torch.empty(batch * n, dtype=shard.dtype, device=shard.device)
    b, e = shard_bounds(batch, world, rank)
    full[b * n:e * n] = shard
    for r in range(world):
        if r == dst:
            continue
        rb, re_ = shard_bounds(batch, world, r)
        if re_ > rb:
            dist.recv(full[rb * n:re_ * n], src=r, group=group)
    return full


def _dev(t):
    return t.device
