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
