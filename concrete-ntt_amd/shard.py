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


def _equal_shards(batch, world):
    return batch % world == 0 and batch > 0


def scatter_batch(full, n, src=0, group=None):
    """Distribute `full` (B*n elements on rank `src`, ignored elsewhere) into per-rank shards.
    Returns this rank's shard tensor.  Equal shards (B divisible by the world size) go through ONE
    `dist.scatter` -- on RCCL that is a single group of ncclSend/ncclRecv from the root, so each of the root's
    xGMI links carries one shard concurrently; ragged batches use point-to-point sends from the root."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    meta = torch.zeros(1, dtype=torch.int64, device=full.device)
    if rank == src:
        meta[0] = full.numel() // n
    dist.broadcast(meta, src=src, group=group)
    batch = int(meta.item())
    b, e = shard_bounds(batch, world, rank)
    mine = torch.empty((e - b) * n, dtype=full.dtype, device=full.device)
    if _equal_shards(batch, world):
        per = (batch // world) * n
        views = [full[r * per:(r + 1) * per] for r in range(world)] if rank == src else None
        dist.scatter(mine, scatter_list=views, src=src, group=group)
        return mine
    if rank == src:
        mine.copy_(full[b * n:e * n])
        reqs = []
        for r in range(world):
            if r == src:
                continue
            rb, re_ = shard_bounds(batch, world, r)
            if re_ > rb:
                reqs.append(dist.isend(full[rb * n:re_ * n], dst=r, group=group))
        for q in reqs:
            q.wait()
        return mine
    if e > b:
        dist.recv(mine, src=src, group=group)
    return mine


def gather_batch(shard, n, batch, dst=0, group=None, out=None):
    """Inverse of scatter_batch: returns the full batch on rank `dst` (written into `out` when given: batch*n
    elements on the shard's device), None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    full = None
    if rank == dst:
        full = out if out is not None else torch.empty(batch * n, dtype=shard.dtype, device=shard.device)
        if full.numel() != batch * n or full.dtype != shard.dtype:
            raise ValueError("out must hold batch*n elements of the shard's dtype")
    if _equal_shards(batch, world):
        per = (batch // world) * n
        views = [full[r * per:(r + 1) * per] for r in range(world)] if rank == dst else None
        dist.gather(shard.contiguous(), gather_list=views, dst=dst, group=group)
        return full
    if rank != dst:
        if shard.numel():
            dist.send(shard.contiguous(), dst=dst, group=group)
        return None
    b, e = shard_bounds(batch, world, rank)
    full[b * n:e * n] = shard
    for r in range(world):
        if r == dst:
            continue
        rb, re_ = shard_bounds(batch, world, r)
        if re_ > rb:
            dist.recv(full[rb * n:re_ * n], src=r, group=group)
    return full
