"""World-size-2 gloo test of the sharding + gather logic (CPU; no kernels involved)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, q):
    import sys
    sys.path.insert(0, REPO)
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import dist as nd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = nd.shard_range(n_rows, rank, world)
    rows = torch.arange(lo, hi, dtype=torch.float32)
    normals = torch.stack([rows, rows * 2, rows * 3], 1)
    expert = (torch.arange(lo, hi) % 7).to(torch.int32)
    probs = torch.stack([rows + e for e in range(7)], 1)
    n, e, p = nd.gather_shards(normals, expert, probs, n_rows)
    n1, e1, p1 = nd.gather_shards(normals, None, None, n_rows)        # single-tower models: normals only
    assert e1 is None and p1 is None and torch.equal(n1, n)
    n2, e2, p2 = nd.gather_shards(normals, expert, probs, n_rows)     # buffers are reused from shape to shape
    assert torch.equal(n2, n) and torch.equal(e2, e) and torch.equal(p2, p)
    q.put((rank, n.numpy(), e.numpy(), p.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_everything():
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import dist as nd
    for n in (0, 1, 7, 100000, 100003):
        for w in (1, 2, 3, 8):
            r = [nd.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


import pytest


@pytest.mark.parametrize("n_rows", [1001, 1000])      # ragged (shards of 500 and 501) and equal shards
def test_gather_two_ranks_gloo(n_rows):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows = np.arange(n_rows, dtype=np.float32)
    for rank, n, e, p in res:
        assert np.array_equal(n, np.stack([rows, rows * 2, rows * 3], 1))
        assert np.array_equal(e, (np.arange(n_rows) % 7).astype(np.int32))      # int32 survives the f32 bit-cast
        assert np.array_equal(p[:, 3], rows + 3)
