"""File seam and data seam on the GPU: the test_n_est_w_experts.py-compatible CLI, the
get_data_loader mirror, and batching invariance of the end-to-end estimator."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dataset_dir(tmp_path_factory):
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import synth
    d = tmp_path_factory.mktemp("pcp")
    names = ["shapeA", "shapeB"]
    for i, (nm, shape, n) in enumerate(zip(names, ("ellipsoid", "torus"), (3000, 2500))):
        pts, nrm = synth.make_cloud(shape, n=n, seed=50 + i, noise=0.00125 * i)
        np.savetxt(str(d / (nm + ".xyz")), pts, fmt="%.9g")
        np.savetxt(str(d / (nm + ".normals")), nrm, fmt="%.9g")
        np.savetxt(str(d / (nm + ".pidx")), np.arange(0, n, 7), fmt="%d")
    (d / "testset.txt").write_text("\n".join(names) + "\n\n")
    return str(d) + os.sep


def test_data_loader_mirror(dataset_dir, gpu_device):
    from nesti_net_amd.provider import get_data_loader
    loader, ds = get_data_loader(dataset_name="testset.txt", batchSize=1000, indir=dataset_dir,
                                 patch_radius=[0.01, 0.03, 0.05], points_per_patch=512, outputs=[],
                                 patch_point_count_std=0, seed=3627473, identical_epochs=False, use_pca=False,
                                 patch_center="point", point_tuple=1, cache_capacity=100, patch_sample_order="full",
                                 workers=0, dataset_type="test", sparse_patches=False, device=gpu_device)
    assert ds.shape_names == ["shapeA", "shapeB"] and ds.shape_patch_count == [3000, 2500]
    assert len(loader) == 6
    batches = list(loader)
    assert [b[0].shape[0] for b in batches] == [1000] * 5 + [500]          # a batch may span shapes; last is short
    assert batches[0][0].shape == (1000, 1536, 3) and batches[0][1].shape == (1000, 3, 3) and batches[0][2].shape == (1000, 3)
    allp = torch.cat([b[0] for b in batches])
    alln = torch.cat([b[2] for b in batches])
    a = ds.get_shape(0).build(0, 3000)
    b = ds.get_shape(1).build(0, 2500)
    assert torch.equal(allp, torch.cat([a[0], b[0]])) and torch.equal(alln, torch.cat([a[1], b[1]]))
    # .npy cache written next to the .xyz like the reference (utils/pcpnet_dataset.py:251)
    assert os.path.exists(os.path.join(dataset_dir, "shapeA.xyz.npy"))
    # sparse patches follow <shape>.pidx
    _, ds2 = get_data_loader(dataset_name="testset.txt", batchSize=64, indir=dataset_dir, patch_radius=[0.01, 0.03, 0.05],
                             points_per_patch=512, sparse_patches=True, device=gpu_device)
    assert ds2.shape_patch_count == [len(range(0, 3000, 7)), len(range(0, 2500, 7))]
    with pytest.raises(ValueError):
        get_data_loader(dataset_name="testset.txt", batchSize=8, indir=dataset_dir, patch_radius=[0.05],
                        points_per_patch=512, use_pca=True)


def test_cli_writes_reference_file_contract(dataset_dir, tmp_path, gpu_device):
    from nesti_net_amd import weights
    from nesti_net_amd.cli import main
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.evaluate import evaluate_set
    from nesti_net_amd.pipeline import NormalEstimator
    from nesti_net_amd.provider import load_xyz
    results = str(tmp_path / "log") + os.sep
    cfg = NestiConfig()
    W = weights.synthetic_weights(cfg)
    os.makedirs(results)
    weights.save(os.path.join(results, "model.nstw"), W, cfg)
    rc = main(["--results_path", results, "--dataset_name", "synth", "--dataset_path", dataset_dir,
               "--testset", "testset.txt", "--sparse_patches", "1", "--batch_size", "128", "--dtype", "bf16"])
    assert rc == 0
    out = os.path.join(results, "synth_results")
    est = NormalEstimator(cfg, W, dtype="bf16", device=gpu_device, batch=300)       # different batching on purpose
    for nm, n in (("shapeA", 3000), ("shapeB", 2500)):
        normals = np.loadtxt(os.path.join(out, nm + ".normals"))
        experts = np.loadtxt(os.path.join(out, nm + ".experts"))
        probs = np.loadtxt(os.path.join(out, nm + ".experts_probs"))
        rows = len(range(0, n, 7))
        assert normals.shape == (rows, 3) and experts.shape == (rows,) and probs.shape == (rows, 7)
        assert np.allclose(probs.sum(1), 1, atol=1e-5) and np.all(experts == np.argmax(probs, 1))
        first = open(os.path.join(out, nm + ".normals")).readline().split()[0]
        assert "e" in first and len(first) >= 22                                    # np.savetxt '%.18e'
        pts = load_xyz(os.path.join(dataset_dir, nm + ".xyz"))
        n2, e2, p2 = est.estimate(pts, pidx=np.arange(0, n, 7))
        assert np.array_equal(e2, experts.astype(np.int32))
        assert np.array_equal(n2.astype(np.float64), normals)                      # batching-invariant, bit for bit
    assert os.path.exists(os.path.join(out, "log.txt"))
    per_shape, avg = evaluate_set(out, dataset_dir, "testset.txt", sparse_patches=True)
    assert set(per_shape) == {"shapeA", "shapeB"} and 0 <= avg["rms"] <= 90 and 0 <= avg["pgp10"] <= 1


def test_cli_default_mode_calibrates_the_gate_per_shape(dataset_dir, tmp_path, gpu_device):
    """--dtype auto = f16x8c for experts_n_est on the 8^3 grid: the gate margin is calibrated on every shape (its counters are that
    shape's, printed to log.txt), two library batches are in flight on two streams, .experts equal the f16x3 mode's and .normals
    equal the f16x8 mode's bit for bit (the cascade is orthogonal to the expert arithmetic; same guard threshold) and stay within
    2.5e-6 cosine of f16x3's (the FP6 cross terms of the experts' tap layers at 8^3 behind the conditioning guard, tests/test_gpu_x8.py)."""
    from nesti_net_amd import weights
    from nesti_net_amd.cli import main
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.pipeline import NormalEstimator
    from nesti_net_amd.provider import load_xyz
    results = str(tmp_path / "log") + os.sep
    cfg = NestiConfig()
    W = weights.synthetic_weights(cfg)
    os.makedirs(results)
    weights.save(os.path.join(results, "model.nstw"), W, cfg)
    rc = main(["--results_path", results, "--dataset_name", "synth", "--dataset_path", dataset_dir, "--testset", "testset.txt"])
    assert rc == 0
    out = os.path.join(results, "synth_results")
    log = open(os.path.join(out, "log.txt")).read()
    assert log.count("gate margin for shape") == 2 and log.count("two-stage gate on shape") == 2
    est = NormalEstimator(cfg, W, dtype="f16x3", device=gpu_device, batch=1000)
    est8 = NormalEstimator(cfg, W, dtype="f16x8", device=gpu_device, batch=1000)
    for nm, n in (("shapeA", 3000), ("shapeB", 2500)):
        normals = np.loadtxt(os.path.join(out, nm + ".normals"))
        experts = np.loadtxt(os.path.join(out, nm + ".experts"))
        probs = np.loadtxt(os.path.join(out, nm + ".experts_probs"))
        assert normals.shape == (n, 3) and experts.shape == (n,) and probs.shape == (n, 7)
        n2, e2, p2 = est.estimate(load_xyz(os.path.join(dataset_dir, nm + ".xyz")))
        # the command line calibrates the conditioning guard on the first 1024 queries of every shape (cli.py): the same here
        from nesti_net_amd.calibrate import calibrate_x8_guard
        cloud8 = est8.prepare(load_xyz(os.path.join(dataset_dir, nm + ".xyz")))
        sp, sn = cloud8.build(0, min(1024, cloud8.patch_count))
        calibrate_x8_guard(est8.net, sp, sn)
        n8, e8, _ = [t.cpu().numpy() for t in est8.run(cloud8)]
        assert np.array_equal(e2, experts.astype(np.int32)) and np.array_equal(e8, e2)
        assert np.array_equal(n8.astype(np.float64), normals)
        cos = (n2.astype(np.float64) * normals).sum(1) / (np.linalg.norm(n2.astype(np.float64), axis=1) * np.linalg.norm(normals, axis=1))
        assert (1 - cos).max() <= 2.5e-6
        assert np.abs(p2 - probs).max() < 0.05
    # --x8_format 8: the same files with the cross terms in FP8 e4m3 -- identical .experts, .normals within the same bar but not the same bits
    results8 = str(tmp_path / "log8") + os.sep
    os.makedirs(results8)
    weights.save(os.path.join(results8, "model.nstw"), W, cfg)
    assert main(["--results_path", results8, "--dataset_name", "synth", "--dataset_path", dataset_dir, "--testset", "testset.txt", "--x8_format", "8"]) == 0
    a = np.loadtxt(os.path.join(out, "shapeA.normals"))
    b = np.loadtxt(os.path.join(results8, "synth_results", "shapeA.normals"))
    assert np.array_equal(np.loadtxt(os.path.join(out, "shapeA.experts")), np.loadtxt(os.path.join(results8, "synth_results", "shapeA.experts")))
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    assert 0 < (1 - cos).max() <= 5e-6 and not np.array_equal(a, b)
    with pytest.raises(SystemExit):
        main(["--results_path", results8, "--dataset_name", "synth", "--dataset_path", dataset_dir, "--testset", "testset.txt", "--dtype", "f16x3", "--x8_format", "6"])


def test_hipgraph_replay_matches_eager(gpu_device):
    """BASELINE config 4 ingredient: the captured forward (hipGraph) replays bit-identically, in f16, over a
    stream of clouds of different density without host synchronisation in between.  The gate is calibrated so that
    the queries spread over all experts: with everything routed to one expert the device-side routing counters
    saturate at the batch size and a replay that fails to reset them goes unnoticed (it did: the hipMemsetAsync node
    that used to clear them was not effective on replay).  Full batches (replays) and ragged tails (eager launches on
    the same workspace) alternate, and every cloud is run twice."""
    from nesti_net_amd import synth, weights
    from nesti_net_amd.calibrate import calibrate_gate
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.pipeline import NormalEstimator
    from nesti_net_amd.provider import CloudPatches
    cfg = NestiConfig()
    W = weights.synthetic_weights(cfg)
    clouds = [synth.make_cloud("sphere", n=6000, seed=5, density="gradient")[0],
              synth.make_cloud("box", n=5000, seed=6, density="striped")[0],
              synth.make_cloud("torus", n=4000, seed=7, noise=0.006)[0]]
    cp = CloudPatches(clouds[2], cfg, device=gpu_device)
    sp, sn = cp.build(0, 512)
    W = calibrate_gate(cfg, W, sp, sn, device=gpu_device)
    eager = NormalEstimator(cfg, W, dtype="f16", device=gpu_device, batch=256)
    graphed = NormalEstimator(cfg, W, dtype="f16", device=gpu_device, batch=256, use_graph=True)
    prepared = [graphed.prepare(c, pidx=np.arange(0, len(c), 9)) for c in clouds]
    outs = [graphed.run(pc) for pc in prepared + prepared]          # enqueued back to back, no sync
    torch.cuda.synchronize()
    seen = set()
    for k, (n_g, e_g, p_g) in enumerate(outs):
        c = clouds[k % 3]
        n_e, e_e, p_e = eager.estimate(c, pidx=np.arange(0, len(c), 9))
        assert np.array_equal(e_g.cpu().numpy(), e_e), "routing differs on run %d" % k
        assert np.array_equal(n_g.cpu().numpy(), n_e) and np.array_equal(p_g.cpu().numpy(), p_e)
        seen |= set(e_e.tolist())
    assert len(seen) >= 5, "the calibrated gate should exercise most experts: %s" % sorted(seen)


def test_two_stream_batches_match_single_stream(gpu_device):
    """Consecutive batches alternating between two HIP streams (own staging buffers and scratch arenas) give the
    single-stream results bit for bit -- with a calibrated gate, so that the per-expert routing lists and device-side
    counters of the two arenas really differ from batch to batch."""
    from nesti_net_amd import synth, weights
    from nesti_net_amd.calibrate import calibrate_gate
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.pipeline import NormalEstimator
    from nesti_net_amd.provider import CloudPatches
    cfg = NestiConfig()
    W = weights.synthetic_weights(cfg)
    pts = synth.make_cloud("ellipsoid", n=5000, seed=3, noise=0.006)[0]
    cp = CloudPatches(pts, cfg, device=gpu_device)
    sp, sn = cp.build(0, 512)
    W = calibrate_gate(cfg, W, sp, sn, device=gpu_device)
    one = NormalEstimator(cfg, W, dtype="bf16", device=gpu_device, batch=300)
    two = NormalEstimator(cfg, W, dtype="bf16", device=gpu_device, batch=300, n_streams=2)
    a = one.estimate(pts, pidx=np.arange(0, 5000, 4))
    b = two.estimate(pts, pidx=np.arange(0, 5000, 4))
    assert len(np.unique(a[1])) >= 5
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_cli_single_scale_model(dataset_dir, tmp_path, gpu_device):
    """--model ss_norm_est (the reference's test_n_est.py path, BASELINE config 0): .normals only."""
    from nesti_net_amd.cli import main
    results = str(tmp_path / "log_ss") + os.sep
    os.makedirs(results)
    rc = main(["--results_path", results, "--model", "ss_norm_est", "--dataset_name", "synth", "--dataset_path", dataset_dir,
               "--testset", "testset.txt", "--sparse_patches", "1", "--synthetic_weights", "--dtype", "bf16"])
    assert rc == 0
    out = os.path.join(results, "synth_results")
    normals = np.loadtxt(os.path.join(out, "shapeA.normals"))
    assert normals.shape == (len(range(0, 3000, 7)), 3) and np.all(np.isfinite(normals))
    assert not os.path.exists(os.path.join(out, "shapeA.experts"))


def test_cli_switching_model(dataset_dir, tmp_path, gpu_device):
    """--model ms_sw_n_est (the reference's test_n_est_w_switching.py path): two radii, .normals only (:158)."""
    from nesti_net_amd.cli import main
    results = str(tmp_path / "log_sw") + os.sep
    os.makedirs(results)
    rc = main(["--results_path", results, "--model", "ms_sw_n_est", "--dataset_name", "synth", "--dataset_path", dataset_dir,
               "--testset", "testset.txt", "--sparse_patches", "1", "--synthetic_weights", "--dtype", "f16"])
    assert rc == 0
    out = os.path.join(results, "synth_results")
    for nm, n in (("shapeA", 3000), ("shapeB", 2500)):
        normals = np.loadtxt(os.path.join(out, nm + ".normals"))
        assert normals.shape == (len(range(0, n, 7)), 3) and np.all(np.isfinite(normals))
        assert not os.path.exists(os.path.join(out, nm + ".experts"))


def test_end_to_end_3_gaussian_grid(gpu_device):
    """--num_gaussians 3 model through the whole path (ball query -> 27-Gaussian MuPS -> conv_net_3g MoE): the f32
    pipeline equals oracle patches -> oracle MuPS -> oracle network on every query of a small cloud subset."""
    from nesti_net_amd import synth, weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.pipeline import NormalEstimator
    from oracle import mups_ref, net_ref, patches_ref
    cfg = NestiConfig(n_gaussians=3, gmm_variance=0.111)
    W = weights.synthetic_weights(cfg)
    pts = synth.make_cloud("torus", n=12000, seed=21, noise=0.006)[0]
    q = np.arange(0, 12000, 1000)
    est = NormalEstimator(cfg, W, dtype="f32", device=gpu_device, batch=5)       # ragged batches: 5 + 5 + 2
    normals, expert, probs = est.estimate(pts, pidx=q)
    _, r_abs = patches_ref.patch_radii(pts, cfg.patch_radius)
    o_pts, o_neff, _, _ = patches_ref.extract_patches(pts, q, r_abs, cfg.num_point, est.seed)
    ref = net_ref.moe_forward(mups_ref.mups_assemble(o_pts, o_neff, 3, grid_n=3, variance=0.111), W, dtype=torch.float64,
                              top1_only=True)
    assert np.array_equal(expert, ref["expert"].numpy())
    a, b = normals.astype(np.float64), ref["normals"].numpy()
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    assert np.all(1 - cos < 1e-5)
    assert np.abs(probs - ref["probs"].numpy()).max() < 2e-5


def test_full_size_cloud_batching_and_sharding_invariance(gpu_device):
    """BASELINE config 2/3 size: a 100k-point cloud through the whole bf16 hot path.  Size-independent properties:
    the result does not depend on how the rows are batched (25 000 vs 16 384 per library call), nor on how they are
    sharded (rows [0, 50k) and [50k, 100k) computed separately, as two ranks would), every output is finite and the
    calibrated gate uses all seven experts."""
    from nesti_net_amd import synth, weights
    from nesti_net_amd.calibrate import calibrate_gate
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.pipeline import NormalEstimator
    from nesti_net_amd.provider import CloudPatches
    cfg = NestiConfig()
    W = weights.synthetic_weights(cfg)
    pts = synth.make_cloud("ellipsoid", n=100000, seed=1234)[0]
    cp = CloudPatches(pts, cfg, device=gpu_device)
    sp, sn = cp.build(0, 512)
    W = calibrate_gate(cfg, W, sp, sn, device=gpu_device)
    del cp
    a = NormalEstimator(cfg, W, dtype="bf16", device=gpu_device, batch=25000)
    cloud = a.prepare(pts)
    na, ea, pa = [t.clone() for t in a.run(cloud)]
    lo = [t.clone() for t in a.run(cloud, 0, 50000)]
    hi = [t.clone() for t in a.run(cloud, 50000, 50000)]
    torch.cuda.synchronize()
    del a
    torch.cuda.empty_cache()
    b = NormalEstimator(cfg, W, dtype="bf16", device=gpu_device, batch=16384)
    nb, eb, pb = b.run(b.prepare(pts))
    torch.cuda.synchronize()
    assert torch.equal(ea, eb) and torch.equal(na, nb) and torch.equal(pa, pb)
    assert torch.equal(torch.cat([lo[0], hi[0]]), na) and torch.equal(torch.cat([lo[1], hi[1]]), ea)
    assert bool(torch.isfinite(na).all()) and bool(torch.isfinite(pa).all())
    hist = torch.bincount(ea.long(), minlength=7)
    assert int((hist > 1000).sum()) == 7, hist.tolist()


def test_config4_stream_of_32_clouds_graph_matches_eager(gpu_device):
    """BASELINE config 4 at its stated size: 32 clouds of 50k-100k points (PCPNet noise levels, gradient / striped
    density sets, the bench's make_clouds(stream=True) recipe) in flight back to back through the f16 hipGraph path --
    full batches replay the captured forward, ragged tails run eagerly on the same workspace -- must equal the eager
    path bit for bit on thirteen of them (every combination of the recipe); and reference-captured fixture rows routed through a graph replay must match the
    fp64 oracle within f16's stated bounds (arg-max unless the oracle's top-2 margin is inside the dtype's probability
    error, normals within 1e-3 cosine; tests/test_gpu_fixtures.py holds the distribution)."""
    import bench
    from conftest import golden_patch_files, load_golden_patches
    from nesti_net_amd import weights
    from nesti_net_amd.calibrate import calibrate_gate
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.pipeline import NormalEstimator
    from nesti_net_amd.provider import CloudPatches
    from oracle import mups_ref, net_ref, patches_ref
    cfg = NestiConfig()
    clouds = [p for p, _ in bench.make_clouds(32, 100000, stream=True)]
    assert min(len(c) for c in clouds) >= 50000 and max(len(c) for c in clouds) <= 100000
    cp = CloudPatches(clouds[0], cfg, device=gpu_device)
    sp, sn = cp.build(0, 512)
    W = calibrate_gate(cfg, weights.synthetic_weights(cfg), sp, sn, device=gpu_device)
    del cp, sp, sn
    B = 8192
    graphed = NormalEstimator(cfg, W, dtype="f16", device=gpu_device, batch=B, use_graph=True)
    prepared = [graphed.prepare(c) for c in clouds]
    outs = [graphed.run(pc) for pc in prepared]                      # 32 clouds enqueued back to back, no host sync
    torch.cuda.synchronize()
    outs = [[t.cpu().numpy() for t in o] for o in outs]
    eager = NormalEstimator(cfg, W, dtype="f16", device=gpu_device, batch=B)
    seen = np.zeros(7, np.int64)
    for k, pc in enumerate(prepared):
        assert np.all(np.isfinite(outs[k][0])) and len(outs[k][0]) == len(clouds[k])
        seen += np.bincount(outs[k][1], minlength=7)
        # the eager twin of the first twelve clouds -- every (shape / noise level, density set) combination of the recipe -- and of the last
        # one (the suite's time budget: the eager pass of all 32 took as long as the graph pass itself)
        if k < 12 or k == len(prepared) - 1:
            n_e, e_e, p_e = [t.cpu().numpy() for t in eager.run(pc)]
            assert np.array_equal(outs[k][1], e_e), "routing differs on cloud %d" % k
            assert np.array_equal(outs[k][0], n_e) and np.array_equal(outs[k][2], p_e), "cloud %d" % k
    assert np.all(seen > 1000), seen
    # fixture rows through a graph replay: one full batch whose first rows are the reference-captured queries
    g = load_golden_patches([p for p in golden_patch_files() if "ellipsoid100k" in p][0])
    fq = g["queries"].astype(np.int64)
    q = np.concatenate([fq, np.arange(B - len(fq)) * 11 % 100000])
    n_g, e_g, p_g = graphed.estimate(g["pts"], pidx=q)
    o_pts, o_neff, _, _ = patches_ref.extract_patches(g["pts"], fq, g["r_abs"], cfg.num_point, g["seed"])
    ref = net_ref.moe_forward(mups_ref.mups_assemble(o_pts, o_neff, 3), W, dtype=torch.float64, top1_only=True)
    srt = np.sort(ref["probs"].numpy(), axis=1)
    margin = srt[:, -1] - srt[:, -2]
    agree = e_g[:len(fq)] == ref["expert"].numpy()
    assert np.all(agree | (margin < 0.06))
    a, b = n_g[:len(fq)][agree].astype(np.float64), ref["normals"].numpy()[agree]
    cos = (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    assert np.all(1 - cos < 1e-3)


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_fused_entry_matches_the_two_call_path(gpu_device, dtype):
    """nesti_estimate_normals (cloud -> patches_mups_kernel -> gate -> routed experts; the patch tensors are never
    written) against the parity entry points it fuses (nesti_patches_query -> nesti_forward, which materialise the
    reference's [B, S*P, 3] placeholder): bit-identical outputs, on a cloud whose small-radius balls hold only a
    handful of points (striped density + the strongest PCPNet noise), with sparse queries and ragged batches."""
    import ctypes
    from nesti_net_amd import _lib, synth, weights
    from nesti_net_amd.calibrate import calibrate_gate
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    from nesti_net_amd.pipeline import NormalEstimator
    from nesti_net_amd.provider import CloudPatches
    cfg = NestiConfig()
    pts = synth.make_cloud("sphere", n=40000, seed=1238, noise=0.012, density="striped")[0]
    q = np.arange(5, 40000, 61)
    cp = CloudPatches(pts, cfg, device=gpu_device, pidx=q)
    p_all, n_all = cp.build(0, len(q))
    assert int(n_all[:, 0].min()) <= 3                      # tiny balls are in the set
    W = calibrate_gate(cfg, weights.synthetic_weights(cfg), p_all, n_all, device=gpu_device)
    net = NestiNet(cfg, W, dtype=dtype, device=gpu_device, max_batch=len(q))
    ref = [t.cpu().numpy() for t in net(p_all, n_all)]
    est = NormalEstimator(cfg, W, dtype=dtype, device=gpu_device, batch=250)       # 656 queries: 250 + 250 + 156
    assert est._fused
    got = est.estimate(pts, pidx=q)
    for x, y in zip(got, ref):
        assert np.array_equal(x, y)
    assert len(np.unique(ref[1])) >= 5
    # misuse is reported, not executed
    lib = _lib.load()
    cloud = est.prepare(pts)
    out = torch.empty((10, 3), dtype=torch.float32, device=gpu_device)
    args = lambda row0, m, ws_bytes: (est.net._handle, _lib.ptr(cloud.cloud), cloud.n_points, None, m, cloud._r, ctypes.c_uint64(1),
                                      row0, 250, 0, _lib.ptr(cloud._ws), cloud._ws.numel(), _lib.ptr(est._arena), ws_bytes,
                                      _lib.ptr(out), None, None, None)
    assert lib.nesti_estimate_normals(*args(39995, 10, est._arena.numel())) != 0 and b"exceed the cloud" in lib.nesti_last_error()
    assert lib.nesti_estimate_normals(*args(0, 10, 1024)) != 0 and b"workspace too small" in lib.nesti_last_error()
    assert lib.nesti_estimate_normals(*args(0, 10, est._arena.numel())) == 0
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())


def test_run_many_equals_per_shape_runs(gpu_device):
    """nesti_estimate_normals_multi: several shapes / shards as ONE stream of batches (batches straddle the items) give
    exactly the per-shape results -- sparse and full query sets, an empty item, batch smaller than an item."""
    from nesti_net_amd import synth, weights
    from nesti_net_amd.calibrate import calibrate_gate
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.pipeline import NormalEstimator
    from nesti_net_amd.provider import CloudPatches
    cfg = NestiConfig()
    clouds = [synth.make_cloud("torus", n=3000, seed=31, noise=0.00125)[0], synth.make_cloud("box", n=2000, seed=32)[0],
              synth.make_cloud("sphere", n=2500, seed=33, density="gradient")[0]]
    cp = CloudPatches(clouds[0], cfg, device=gpu_device)
    sp, sn = cp.build(0, 512)
    W = calibrate_gate(cfg, weights.synthetic_weights(cfg), sp, sn, device=gpu_device)
    est = NormalEstimator(cfg, W, dtype="f16", device=gpu_device, batch=700)
    prepared = [est.prepare(clouds[0]), est.prepare(clouds[1], pidx=np.arange(0, 2000, 3)), est.prepare(clouds[2])]
    items = [(prepared[0], 100, 1500), (prepared[1], 0, prepared[1].patch_count), (prepared[2], 0, 0), (prepared[2], 2000, 500)]
    many = est.run_many(items)
    torch.cuda.synchronize()
    assert [m[0].shape[0] for m in many] == [1500, prepared[1].patch_count, 0, 500]
    for (c, f, n), got in zip(items, many):
        ref = est.run(c, f, n)
        for x, y in zip(got, ref):
            assert torch.equal(x, y)


@pytest.mark.parametrize("mode", ["reference", "reference_host"])
def test_reference_order_subsample_feeds_the_same_forward(gpu_device, mode):
    """NormalEstimator(subsample='reference' / 'reference_host'): the reference's own thinning of balls larger than P (cKDTree
    visiting order + one RandomState stream in visiting order, utils/pcpnet_dataset.py:304-321; 'reference' = on the GPU from the
    natively replayed stream, 'reference_host' = scipy + numpy on the host) in front of the usual forward pass.  On the
    100k golden fixture -- 32 queries, the largest scale capped on every one -- the outputs equal a forward pass over
    the patch tensors the reference dataset itself produced, bit for bit, and a second cloud continues the stream."""
    from conftest import golden_patch_files, load_golden_patches
    from nesti_net_amd import weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    from nesti_net_amd.pipeline import NormalEstimator
    g = load_golden_patches([p for p in golden_patch_files() if "ellipsoid100k" in p][0])
    cfg = NestiConfig()
    W = weights.synthetic_weights(cfg)
    net = NestiNet(cfg, W, dtype="f16x3", device=gpu_device, max_batch=len(g["queries"]))
    want = [t.cpu().numpy() for t in net(torch.as_tensor(g["points"], device=gpu_device), torch.as_tensor(g["n_eff"], device=gpu_device))]
    est = NormalEstimator(cfg, W, dtype="f16x3", device=gpu_device, batch=10, seed=g["seed"], subsample=mode)
    got = est.estimate(g["pts"], pidx=g["queries"])
    assert (g["n_eff"] == cfg.num_point).any()
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    again = est.estimate(g["pts"], pidx=g["queries"])           # the stream has moved on: capped rows are thinned differently
    assert not np.array_equal(again[0], got[0])
    with pytest.raises(ValueError):
        NormalEstimator(cfg, W, dtype="f16", device=gpu_device, batch=4, subsample="sorted")


@pytest.mark.parametrize("dtype", ["f16", "f16x3"])
def test_later_expert_rounds_run_through_the_walking_kernels(gpu_device, dtype):
    """A batch larger than 8192 sizes the expert towers for a quarter of it and walks each routing list in four rounds; rounds
    1 .. 3 are normally empty and are launched as small walking grids (csrc/kernels.h: ConvParams::walk).  With a gate biased towards
    one expert most queries route to it, so that expert's later rounds are FULL: the walking instantiations of conv_igemm /
    conv8n / conv4n then do real work, and the result must equal the small-batch (single-round) result bit for bit."""
    from nesti_net_amd import synth, weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.pipeline import NormalEstimator
    cfg = NestiConfig()
    W = dict(weights.synthetic_weights(cfg))
    b4 = np.array(W["fc4noise/biases"], dtype=np.float32, copy=True)
    b4[3] += 1000.0                                         # a gate that strongly prefers expert 3 (fc4 has no batch norm)
    W["fc4noise/biases"] = b4
    pts = synth.make_cloud("torus", n=30000, seed=5)[0]
    q = np.arange(12000)
    big = NormalEstimator(cfg, W, dtype=dtype, device=gpu_device, batch=12000).estimate(pts, pidx=q)
    small = NormalEstimator(cfg, W, dtype=dtype, device=gpu_device, batch=3000).estimate(pts, pidx=q)
    hist = np.bincount(big[1], minlength=cfg.n_experts)
    assert hist.max() > 2 * 3072, hist.tolist()             # the favourite expert needs at least three of its four rounds
    for a, b in zip(big, small):
        assert np.array_equal(a, b)
    assert np.all(np.isfinite(big[0]))
