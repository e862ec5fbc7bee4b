"""Multi-GPU execution: one process per GPU, ``torch.distributed`` (backend 'nccl' = RCCL over
xGMI on ROCm, 'gloo' in CPU tests).

Query points are independent at inference (BN uses stored EMA statistics,
``utils/tf_util.py:491-493``), so a shape's patch rows are block-partitioned across ranks --
contiguous ranges keep output order = file order -- the cloud and the weights are replicated,
and the only exchange is ONE all-gather of the per-shard results per shape
(normals 3 + expert 1 + probs E floats per point, ~4.4 MB for 100k points).

dtypes 'f16x3c' / 'f16x8c': the gather buffer carries one spare row per rank holding that rank's ``max_margin_err`` (the
largest error its f16 gate filter has shown); after the gather every rank folds all of them into its own counter
(``NestiNet.export_gate_error`` / ``import_gate_error``: two one-thread kernels, no host synchronisation, no extra
collective), so from the next step on ALL ranks filter with the same ``tau_eff`` = 1.5 x the largest error any rank has
measured.  Scope of that guarantee: the imported maximum lives in the same device counter as the rank's own
measurements, so (a) it lasts until the counters are reset -- ``calibrate_gate_margin`` resets them for every shape, after which
the ranks re-converge with the next gather -- and (b) ``cascade_stats()['max_margin_err']`` of a rank reads the largest error ANY
rank has measured since the last reset, not that rank's alone.  The export runs on the caller's current stream after
``NormalEstimator.run`` / ``run_many`` have joined their lane streams (they end with ``main.wait_stream(lane)``)."""
import torch
import torch.distributed as dist


def shard_range(n_rows, rank, world):
    """Contiguous block of rows for ``rank``: [lo, hi)."""
    lo = (n_rows * rank) // world
    hi = (n_rows * (rank + 1)) // world
    return lo, hi


def max_shard(n_rows, world):
    return max(shard_range(n_rows, r, world)[1] - shard_range(n_rows, r, world)[0] for r in range(world))


def agree_on_gate_margin(net, tau, device, group=None):
    """dtype 'f16x3c': every rank calibrates its gate margin on the same sample, but sigma comes out of floating-point atomics
    whose order differs from run to run, so the ranks' values may differ in their last bits.  All ranks adopt the largest one
    (one scalar all-reduce, outside any timed region).  Returns (tau everyone uses, max - min over the ranks)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return min(float(tau), 1e30), 0.0
    big = 1e30                                     # calibrate_gate_margin's "no filtering" value (tau = inf is not JSON)
    tau = min(float(tau), big)
    t = torch.tensor([tau, -tau], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, lo = float(t[0].item()), -float(t[1].item())
    net.set_gate_margin(hi)
    return hi, (hi - lo if hi < big else 0.0)


_BUFFERS = {}


def _buffers(world, rows, cols, device):
    """Preallocated send [rows + 1, cols] / receive [world, rows + 1, cols] f32 buffers, reused from shape to shape.  Row
    ``rows`` is the spare row: element 0 carries the rank's gate error (dtype 'f16x3c'), zero otherwise."""
    key = (world, rows, cols, str(device))
    if key not in _BUFFERS:
        _BUFFERS[key] = (torch.zeros((rows + 1, cols), dtype=torch.float32, device=device),
                         torch.empty((world, rows + 1, cols), dtype=torch.float32, device=device))
    return _BUFFERS[key]


def _cascade_net(estimator):
    """The estimator's model when it runs the two-stage gate (its error counter takes part in the gather), else None."""
    net = getattr(estimator, "net", None)
    return net if getattr(net, "cascade", False) else None


def _all_gather(mine, everyone, rows, net, group):
    """THE collective of the path; ``net`` (a cascade model or None): its gate error rides in the spare row."""
    world, _, cols = everyone.shape
    if net is not None:
        net.export_gate_error(mine[rows, 0:1])
    dist.all_gather_into_tensor(everyone.view(world * (rows + 1), cols), mine, group=group)
    if net is not None:
        net.import_gate_error(everyone[:, rows, 0].contiguous())


def pack_results(normals, expert, probs, out):
    """[n,3] f32 (+ [n] int32 bit-cast + [n,E] f32 when the model has a gate) -> rows [0, n) of ``out``."""
    n = normals.shape[0]
    out[:n, 0:3] = normals
    if expert is not None:
        out[:n, 3] = expert.to(torch.int32).view(torch.float32)
        out[:n, 4:] = probs
    return out


def unpack_results(buf, gated=True):
    if not gated:
        return buf[:, 0:3].clone(), None, None      # [.., 3] is the whole row here: contiguous() would alias the cached buffer
    return buf[:, 0:3].contiguous(), buf[:, 3].contiguous().view(torch.int32), buf[:, 4:].contiguous()


def gather_shards(normals, expert, probs, n_rows, group=None, net=None):
    """All-gather the per-rank shard results of one shape into full-length tensors on every rank: ONE
    ``all_gather_into_tensor`` on a preallocated [world, max_shard, 3 (+1+E)] buffer (RCCL ring over xGMI; ~4.4 MB
    for a 100k-point cloud, latency-bound).  Single-tower models (ss_norm_est / ms_norm_est) have no expert/probs
    columns."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return normals, expert, probs
    world = dist.get_world_size(group)
    gated = expert is not None
    cols = 3 + ((1 + probs.shape[1]) if gated else 0)
    ms = max_shard(n_rows, world)
    mine, everyone = _buffers(world, ms, cols, normals.device)
    pack_results(normals, expert, probs, mine)
    _all_gather(mine, everyone, ms, net, group)                                       # the single collective of the path
    outs = []
    for r in range(world):
        lo, hi = shard_range(n_rows, r, world)
        outs.append(everyone[r, :hi - lo])
    return unpack_results(torch.cat(outs), gated)


def estimate_sharded(estimator, cloud, group=None):
    """Run this rank's block of ``cloud``'s patch rows and gather everyone's results."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(cloud.patch_count, rank, world)
    normals, expert, probs = estimator.run(cloud, lo, hi - lo)
    return gather_shards(normals, expert, probs, cloud.patch_count, group, net=_cascade_net(estimator))


def estimate_sharded_many(estimator, clouds, group=None):
    """All of ``clouds`` in one go: this rank's row blocks of every cloud form ONE stream of library batches
    (``NormalEstimator.run_many``: at 8 ranks a 100k cloud leaves 12.5k rows per rank -- eight such blocks fill the gate
    and expert launches like one whole cloud does) and ONE all-gather carries every cloud's results.  Returns one
    (normals, expert, probs) triple of full-length tensors per cloud, on every rank."""
    inited = dist.is_initialized()
    world = dist.get_world_size(group) if inited else 1
    rank = dist.get_rank(group) if inited else 0
    ranges = [shard_range(c.patch_count, rank, world) for c in clouds]
    outs = estimator.run_many([(c, lo, hi - lo) for c, (lo, hi) in zip(clouds, ranges)])
    if world == 1:
        return outs
    gated = outs[0][1] is not None
    cols = 3 + ((1 + outs[0][2].shape[1]) if gated else 0)
    ms = [max_shard(c.patch_count, world) for c in clouds]
    offs = [0]
    for m in ms:
        offs.append(offs[-1] + m)
    mine, everyone = _buffers(world, offs[-1], cols, outs[0][0].device)
    for (n, e, p), o in zip(outs, offs):
        pack_results(n, e, p, mine[o:])
    _all_gather(mine, everyone, offs[-1], _cascade_net(estimator), group)                     # the single collective of the step
    res = []
    for ci, c in enumerate(clouds):
        parts = []
        for r in range(world):
            lo, hi = shard_range(c.patch_count, r, world)
            parts.append(everyone[r, offs[ci]:offs[ci] + hi - lo])
        res.append(unpack_results(torch.cat(parts), gated))
    return res
