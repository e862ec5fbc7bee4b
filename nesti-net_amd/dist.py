"""Multi-GPU execution: one process per GPU, ``torch.distributed`` (backend 'nccl' = RCCL over
xGMI on ROCm, 'gloo' in CPU tests).

Query points are independent at inference (BN uses stored EMA statistics,
``utils/tf_util.py:491-493``), so a shape's patch rows are block-partitioned across ranks --
contiguous ranges keep output order = file order -- the cloud and the weights are replicated,
and the only exchange is ONE all-gather of the per-shard results per shape
(normals 3 + expert 1 + probs E floats per point, ~4.4 MB for 100k points)."""
import torch
import torch.distributed as dist


def shard_range(n_rows, rank, world):
    """Contiguous block of rows for ``rank``: [lo, hi)."""
    lo = (n_rows * rank) // world
    hi = (n_rows * (rank + 1)) // world
    return lo, hi


def max_shard(n_rows, world):
    return max(shard_range(n_rows, r, world)[1] - shard_range(n_rows, r, world)[0] for r in range(world))


def pack_results(normals, expert, probs, pad_to):
    """[n,3] f32, [n] int32, [n,E] f32 -> one [pad_to, 4+E] f32 buffer (expert bit-cast)."""
    n, E = normals.shape[0], probs.shape[1]
    buf = torch.zeros((pad_to, 4 + E), dtype=torch.float32, device=normals.device)
    buf[:n, 0:3] = normals
    buf[:n, 3] = expert.to(torch.int32).view(torch.float32)
    buf[:n, 4:] = probs
    return buf


def unpack_results(buf):
    return buf[:, 0:3].contiguous(), buf[:, 3].contiguous().view(torch.int32), buf[:, 4:].contiguous()


def gather_shards(normals, expert, probs, n_rows, group=None):
    """All-gather the per-rank shard results of one shape into full-length tensors on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if not dist.is_initialized():
        return normals, expert, probs
    ms = max_shard(n_rows, world)
    mine = pack_results(normals, expert, probs, ms)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)       # the single collective of the path
    outs = []
    for r in range(world):
        lo, hi = shard_range(n_rows, r, world)
        outs.append(parts[r][:hi - lo])
    return unpack_results(torch.cat(outs))


def estimate_sharded(estimator, cloud, group=None):
    """Run this rank's block of ``cloud``'s patch rows and gather everyone's results."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(cloud.patch_count, rank, world)
    normals, expert, probs = estimator.run(cloud, lo, hi - lo)
    return gather_shards(normals, expert, probs, cloud.patch_count, group)
