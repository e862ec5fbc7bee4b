"""World-size-2 gloo test of the sharding + gather logic (CPU; no kernels involved)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import pytest
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


class _FakeCloud:
    def __init__(self, n):
        self.patch_count = n


class _FakeEstimator:
    """run_many of a deterministic function of (cloud size, row): exercises estimate_sharded_many without a GPU."""

    def run_many(self, items):
        out = []
        for c, first, count in items:
            rows = torch.arange(first, first + count, dtype=torch.float32) + 1000.0 * c.patch_count
            out.append((torch.stack([rows, rows * 2, rows * 3], 1), (torch.arange(first, first + count) % 7).to(torch.int32),
                        torch.stack([rows + e for e in range(7)], 1)))
        return out


def _worker_many(rank, world, port, sizes, q):
    import sys
    sys.path.insert(0, REPO)
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import dist as nd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = nd.estimate_sharded_many(_FakeEstimator(), [_FakeCloud(n) for n in sizes])
    res2 = nd.estimate_sharded_many(_FakeEstimator(), [_FakeCloud(n) for n in sizes])      # buffers reused
    assert all(torch.equal(a[0], b[0]) for a, b in zip(res, res2))
    q.put((rank, [(n.numpy(), e.numpy(), p.numpy()) for n, e, p in res]))
    dist.barrier()
    dist.destroy_process_group()


class _FakeNet:
    def __init__(self):
        self.tau = None

    def set_gate_margin(self, tau):
        self.tau = tau


def _worker_margin(rank, world, port, q):
    import sys
    sys.path.insert(0, REPO)
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import dist as nd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _FakeNet()
    tau, spread = nd.agree_on_gate_margin(net, 0.147 + 1e-9 * rank, torch.device("cpu"))
    q.put((rank, tau, spread, net.tau))
    dist.barrier()
    dist.destroy_process_group()


def test_every_rank_adopts_the_same_gate_margin_world_8():
    """The ranks' calibrated margins may differ in their last bits (atomics); all of them must end up filtering with ONE value."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_margin, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    taus = {t for _, t, _, _ in res}
    assert len(taus) == 1 and abs(taus.pop() - (0.147 + 7e-9)) < 1e-12
    assert all(abs(s - 7e-9) < 1e-12 and nt == t for _, t, s, nt in res)


class _FakeCascadeNet:
    """Stands in for an 'f16x3c' NestiNet: one float of gate error out, the other ranks' floats in."""
    cascade = True

    def __init__(self, err):
        self.err = float(err)

    def export_gate_error(self, dst):
        dst[0] = self.err

    def import_gate_error(self, src):
        assert src.is_contiguous()
        v = src[torch.isfinite(src)]
        self.err = max(self.err, float(v.max())) if len(v) else self.err


class _FakeCascadeEstimator(_FakeEstimator):
    def __init__(self, err):
        self.net = _FakeCascadeNet(err)

    def run(self, cloud, first, count):
        return self.run_many([(cloud, first, count)])[0]


def _worker_gate_error(rank, world, port, q):
    import sys
    sys.path.insert(0, REPO)
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import dist as nd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    est = _FakeCascadeEstimator(0.05 + 0.01 * ((rank * 5) % world))          # every rank has measured a different error
    res = nd.estimate_sharded_many(est, [_FakeCloud(1001), _FakeCloud(3)])
    after_many = est.net.err
    est.net.err = 0.2 if rank == 1 else 0.0                                   # a widening event on ONE rank ...
    n, e, p = nd.estimate_sharded(est, _FakeCloud(77))                        # ... reaches everyone with the next gather
    rows = np.arange(77, dtype=np.float32) + 77000.0
    ok = np.array_equal(n.numpy(), np.stack([rows, rows * 2, rows * 3], 1)) and len(res[0][0]) == 1001
    q.put((rank, after_many, est.net.err, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_gate_error_rides_in_the_gather(world):
    """after a widening event all ranks filter with ONE tau_eff -- the largest f16-gate error any
    rank has measured travels in a spare row of the step's single all-gather (no collective of its own)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_gate_error, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, after_many, after_one, ok in res:
        assert ok
        assert abs(after_many - (0.05 + 0.01 * (world - 1))) < 1e-6, (rank, after_many)
        assert abs(after_one - 0.2) < 1e-6, (rank, after_one)


def test_gate_margin_at_infinity_is_json_safe():
    """fewer than 256 calibration queries leave tau at infinity; the agreed value and the spread must stay finite."""
    import json
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import dist as nd
    net = _FakeNet()
    tau, spread = nd.agree_on_gate_margin(net, float("inf"), torch.device("cpu"))
    assert tau == 1e30 and spread == 0.0
    json.loads(json.dumps({"tau": tau, "spread": spread}, allow_nan=False))


@pytest.mark.parametrize("sizes,world", [([1001, 64, 500], 2),
                                         # world size 8 (the north star's node): sizes not divisible by 8, a cloud with fewer
                                         # rows than ranks (three ranks get nothing of it), an empty cloud, equal shards
                                         ([100003, 5, 12501, 0, 64], 8)])
def test_sharded_many_gloo(sizes, world):
    """One all-gather for several clouds of different (ragged) sizes: every rank ends up with every cloud in row order."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_many, args=(r, world, port, sizes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == list(range(world))
    for rank, clouds in res:
        for n_rows, (n, e, p) in zip(sizes, clouds):
            rows = np.arange(n_rows, dtype=np.float32) + 1000.0 * n_rows
            assert np.array_equal(n, np.stack([rows, rows * 2, rows * 3], 1))
            assert np.array_equal(e, (np.arange(n_rows) % 7).astype(np.int32))
            assert np.array_equal(p[:, 6], rows + 6)


def test_shard_ranges_cover_everything():
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import dist as nd
    for n in (0, 1, 7, 100000, 100003):
        for w in (1, 2, 3, 8):
            r = [nd.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


@pytest.mark.parametrize("n_rows,world", [(1001, 2), (1000, 2),      # ragged (shards of 500 and 501) and equal shards
                                          (100003, 8), (5, 8)])     # the 8-GPU node: ragged, and fewer rows than ranks
def test_gather_gloo(n_rows, world):
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


def _worker_alias(rank, world, port, q):
    import sys
    sys.path.insert(0, REPO)
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import dist as nd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_rows = 1000                                         # equal shards: the receive buffer itself is handed back
    lo, hi = nd.shard_range(n_rows, rank, world)
    first = torch.arange(lo, hi, dtype=torch.float32)[:, None] * torch.tensor([1.0, 2.0, 3.0])
    a, _, _ = nd.gather_shards(first, None, None, n_rows)            # single-tower model: normals only
    keep = a.clone()
    b, _, _ = nd.gather_shards(first + 7.0, None, None, n_rows)      # same shape, different data, same cached buffers
    q.put((rank, bool(torch.equal(a, keep)), bool(torch.equal(b, keep + 7.0))))
    dist.barrier()
    dist.destroy_process_group()


def test_single_tower_gather_does_not_alias_the_cached_buffer():
    """with no expert / probs columns the [.., 0:3] slice of the receive buffer is already contiguous, so
    .contiguous() returned a view of the cached buffer and the next gather of the same shape overwrote results the caller
    still held."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_alias, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, first_unchanged, second_right in res:
        assert first_unchanged and second_right, rank
