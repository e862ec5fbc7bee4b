"""HIP patch extraction (nesti_patches_grid/query through the C-ABI) vs oracle/patches_ref.py
(bit-exact) and vs the patches captured from the reference's PointcloudPatchDataset."""
import numpy as np
import pytest
import torch

from conftest import golden_patch_files, load_golden_patches

pytestmark = pytest.mark.gpu


def _cfg(g):
    from nesti_net_amd.config import NestiConfig
    S = len(g["radii"])
    return NestiConfig(patch_radius=[float(r) for r in g["radii"]], num_point=g["P"], n_experts=1,
                       expert_dict={0: list(range(S))})


@pytest.mark.parametrize("path", golden_patch_files(), ids=lambda p: p.split("patches_")[-1][:-4])
def test_patches_bit_exact_vs_oracle_and_reference(path, gpu_device):
    from nesti_net_amd.provider import CloudPatches
    from oracle import patches_ref
    g = load_golden_patches(path)
    cfg = _cfg(g)
    cp = CloudPatches(g["pts"], cfg, device=gpu_device, seed=g["seed"], pidx=g["queries"])
    assert np.array_equal(np.asarray(cp.r_abs), g["r_abs"])
    M = len(g["queries"])
    points, n_eff, nbr, n_ball = [t.cpu().numpy() for t in cp.build(0, M, want_idx=True)]
    o_points, o_n_eff, o_nbr, o_n_ball = patches_ref.extract_patches(g["pts"], g["queries"], cp.r_abs, g["P"], g["seed"])
    assert np.array_equal(n_ball, o_n_ball)
    assert np.array_equal(n_eff, o_n_eff) and np.array_equal(n_eff, g["n_eff"])     # == the reference's
    assert np.array_equal(nbr, o_nbr)
    assert np.array_equal(points.view(np.uint32), o_points.view(np.uint32))         # bit-exact
    S, P = n_eff.shape[1], g["P"]
    for q in range(M):
        for s in range(S):
            assert n_ball[q, s] == len(g["balls"][q][s])
            if n_ball[q, s] <= P:
                assert np.array_equal(np.sort(nbr[q, s * P:s * P + n_eff[q, s]]), g["balls"][q][s])


def test_patches_batching_invariance_and_dense_queries(gpu_device):
    """All queries of a cloud, in two different batchings; integer outputs identical; ball sizes
    match scipy for every query (the 'full' sampler path)."""
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.provider import CloudPatches
    from nesti_net_amd import synth
    from oracle import patches_ref
    pts, _ = synth.make_cloud("torus", n=20000, seed=77, noise=0.00125)
    cfg = NestiConfig()
    cp = CloudPatches(pts, cfg, device=gpu_device)
    a = [t.cpu().numpy() for t in cp.build(0, 20000, want_idx=True)]
    parts = [cp.build(f, c, want_idx=True) for f, c in ((0, 7000), (7000, 1), (7001, 12999))]
    b = [torch.cat([p[i] for p in parts]).cpu().numpy() for i in range(4)]
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
    tree = patches_ref.build_tree(pts)
    _, r_abs = patches_ref.patch_radii(pts, cfg.patch_radius)
    for s, r in enumerate(r_abs):
        cnt = tree.query_ball_point(pts[::37], r, return_length=True)
        assert np.array_equal(a[3][::37, s], cnt)
    # every selected neighbour really is inside the ball (fp64 test) and rows are sorted by key
    d = np.linalg.norm(pts[a[2][:50, 2 * 512:2 * 512 + 5]].astype(np.float64) - pts[:50, None, :].astype(np.float64), axis=2)
    assert np.all(d <= r_abs[2] * (1 + 1e-12))


@pytest.mark.parametrize("shape,noise,density", [
    ("sphere", 0.0, None), ("ellipsoid", 0.00125, None), ("torus", 0.006, None), ("box", 0.012, None),   # BASELINE cfg 3: PCPNet noise levels
    ("sphere", 0.0, "gradient"), ("box", 0.0, "striped"), ("torus", 0.00125, "gradient")])             # cfg 4: varying-density sets
def test_patches_pcpnet_noise_levels_and_density_sets(shape, noise, density, gpu_device):
    """Ball sets / n_eff / neighbour order / patch coordinates bit-exact against the (reference-pinned) oracle on
    the PCPNet noise levels and the gradient / striped density sets, where ball sizes swing between a handful
    of points and far more than P."""
    from nesti_net_amd import synth
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.provider import CloudPatches
    from oracle import patches_ref
    pts, _ = synth.make_cloud(shape, n=100000, seed=11, noise=noise, density=density)
    cfg = NestiConfig()
    q = np.arange(0, 100000, 773)
    cp = CloudPatches(pts, cfg, device=gpu_device, pidx=q)
    got = [t.cpu().numpy() for t in cp.build(0, len(q), want_idx=True)]
    ref = patches_ref.extract_patches(pts, q, cp.r_abs, cfg.num_point, cp.seed)
    assert np.array_equal(got[3], ref[3]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    assert np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32))
    assert got[3][:, 2].max() > cfg.num_point          # the subsample branch is exercised
    if density is not None:
        assert got[3][:, 2].max() > 4 * max(1, got[3][:, 2].min())   # density really varies


def test_patches_tiny_and_degenerate_clouds(gpu_device):
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.provider import CloudPatches
    from oracle import patches_ref
    cfg = NestiConfig(num_point=8)
    rng = np.random.RandomState(3)
    for pts in (rng.rand(1, 3).astype(np.float32) + 1,                         # single point: bbdiag = 0 -> r = 0
                rng.rand(5, 3).astype(np.float32),
                np.concatenate([rng.rand(300, 2), np.zeros((300, 1))], 1).astype(np.float32)):   # planar
        bbdiag, r_abs = patches_ref.patch_radii(pts, cfg.patch_radius)
        if bbdiag == 0:
            with pytest.raises(Exception):
                CloudPatches(pts, cfg, device=gpu_device)
            continue
        cp = CloudPatches(pts, cfg, device=gpu_device)
        got = [t.cpu().numpy() for t in cp.build(0, len(pts), want_idx=True)]
        ref = patches_ref.extract_patches(pts, np.arange(len(pts)), r_abs, 8, cp.seed)
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
        assert np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32))


@pytest.mark.parametrize("path", golden_patch_files(), ids=lambda p: p.split("patches_")[-1][:-4])
def test_reference_order_rows_on_the_gpu_equal_the_reference_datasets(path, gpu_device):
    """the reference's own subsample ORDER on the GPU -- ball sizes from the count kernel, the shared
    RandomState replayed natively from those sizes (csrc/refreplay.cpp), the balls sorted into cKDTree's visiting order and the
    picks applied by patches_ref_kernel.  The patch ROWS must equal what the reference's PointcloudPatchDataset itself produced
    (tests/golden, queries visited in order with its seed), bit for bit, the capped balls INCLUDED."""
    from nesti_net_amd.provider import CloudPatches
    from nesti_net_amd.refsample import RefStream
    g = load_golden_patches(path)
    cfg = _cfg(g)
    cp = CloudPatches(g["pts"], cfg, device=gpu_device, seed=g["seed"], pidx=g["queries"])
    M, S, P = len(g["queries"]), len(g["radii"]), g["P"]
    sizes = cp.count_balls(0, M).cpu().numpy()
    assert np.array_equal(sizes, np.array([[len(b) for b in balls] for balls in g["balls"]]))
    stream = RefStream(g["seed"])
    # two batches: the stream carries over, the pick offsets are per call
    cut = M // 3
    got_p, got_n, got_i = [], [], []
    for first, count in ((0, cut), (cut, M - cut)):
        picks, offs = stream.picks(sizes[first:first + count].ravel(), P)
        pk = torch.from_numpy(picks.view(np.int16).copy()).to(gpu_device)
        of = torch.from_numpy(offs.copy()).to(gpu_device)
        p, n, nbr = cp.build_reference_order(first, count, pk, of, want_idx=True)
        got_p.append(p.cpu().numpy()); got_n.append(n.cpu().numpy()); got_i.append(nbr.cpu().numpy())
    points, n_eff, nbr = np.concatenate(got_p), np.concatenate(got_n), np.concatenate(got_i)
    assert np.array_equal(n_eff, g["n_eff"])
    capped = int((sizes > P).sum())
    assert np.array_equal(points.view(np.uint32), g["points"].view(np.uint32)), "%d capped balls" % capped
    if "100k" in path or "gradient" in path:
        assert capped > 0
    for q in range(M):                       # the neighbour indices are members of the ball, without repetition
        for s in range(S):
            sel = nbr[q, s * P:s * P + n_eff[q, s]]
            assert len(np.unique(sel)) == len(sel) and np.isin(sel, g["balls"][q][s]).all()
            assert np.all(nbr[q, s * P + n_eff[q, s]:(s + 1) * P] == -1)


def test_reference_order_gpu_equals_the_host_sampler_on_a_dense_cloud(gpu_device):
    """3 000 queries of a 100k-point cloud with a density gradient (the largest scale over-full on most queries, the middle one on
    the dense side): the GPU reference-order rows equal refsample.ReferencePatchSampler's (scipy + numpy on the host) bit for bit, and
    the estimator's two reference modes write the same normals."""
    from nesti_net_amd import synth, weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.pipeline import NormalEstimator
    from nesti_net_amd.provider import CloudPatches
    from nesti_net_amd.refsample import ReferencePatchSampler, RefStream
    cfg = NestiConfig()
    pts = synth.make_cloud("ellipsoid", n=100000, seed=99, density="gradient")[0]
    q = np.arange(0, 100000, 33)[:3000]
    cp = CloudPatches(pts, cfg, device=gpu_device, pidx=q)
    sizes = cp.count_balls(0, len(q)).cpu().numpy()
    over = (sizes > cfg.num_point).mean(0)
    print("ball sizes: max", sizes.max(0), "mean", sizes.mean(0).round(1), "over-full fraction per scale", over.round(3))
    assert over[2] > 0.5 and (sizes <= cfg.num_point).any()      # both branches of utils/pcpnet_dataset.py:310-321 are exercised
    stream = RefStream(3627473)
    picks, offs = stream.picks(sizes.ravel(), cfg.num_point)
    p, n = cp.build_reference_order(0, len(q), torch.from_numpy(picks.view(np.int16).copy()).to(gpu_device), torch.from_numpy(offs.copy()).to(gpu_device))
    host = ReferencePatchSampler(3627473)
    hp, hn = host.patches(pts, host.build_tree(pts), q.astype(np.int64), cp.r_abs, cfg.num_point)
    assert np.array_equal(n.cpu().numpy(), hn)
    assert np.array_equal(p.cpu().numpy().view(np.uint32), hp.view(np.uint32))
    W = weights.synthetic_weights(cfg)
    a = NormalEstimator(cfg, W, dtype="f16", device=gpu_device, batch=1024, subsample="reference").estimate(pts, pidx=q)
    b = NormalEstimator(cfg, W, dtype="f16", device=gpu_device, batch=700, subsample="reference_host").estimate(pts, pidx=q)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_reference_order_falls_back_to_the_host_for_balls_beyond_the_lds_sort(gpu_device):
    """A shape whose balls exceed the kernel's LDS sort (16 384 points) goes through the host sampler instead -- both draw from ONE
    native stream object, so a normal shape AFTER it still lines up with the all-host run."""
    from nesti_net_amd import _lib, synth, weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.pipeline import NormalEstimator
    from nesti_net_amd.provider import CloudPatches
    cfg = NestiConfig()
    rs = np.random.RandomState(3)
    blob = (rs.randn(30000, 3) * 0.01).astype(np.float32)                      # 30k points in a tiny blob ...
    far = np.array([[40.0, 0, 0], [-40.0, 0, 0]], np.float32)                  # ... and two outliers that stretch the bounding box
    dense = np.concatenate([blob, far])
    q = np.arange(0, 30000, 500)[:48]
    cp = CloudPatches(dense, cfg, device=gpu_device, pidx=q)
    sizes = cp.count_balls(0, len(q)).cpu().numpy()
    assert sizes.max() > _lib.load().nesti_patches_ref_max_ball()
    normal = synth.make_cloud("torus", n=20000, seed=8)[0]
    qn = np.arange(0, 20000, 100)[:200]
    W = weights.synthetic_weights(cfg)
    outs = {}
    for mode in ("reference", "reference_host"):
        est = NormalEstimator(cfg, W, dtype="f16", device=gpu_device, batch=32, subsample=mode)
        outs[mode] = (est.estimate(dense, pidx=q), est.estimate(normal, pidx=qn))
    for a, b in zip(outs["reference"], outs["reference_host"]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
