"""Gating + expert towers (MFMA implicit-GEMM conv stack) through the C-ABI vs the torch-CPU
fp64 oracle (oracle/net_ref.py) with seeded synthetic weights.

Tolerances (north_star): expert arg-max bit-exact, normals within 1e-5 cosine -- asserted in
the exact-fp32 mode ('f32': v_mfma_f32_32x32x2_f32).  The bf16/f16 production modes are
checked against looser, stated bounds."""
import numpy as np
import pytest
import torch

from conftest import golden_patch_files, load_golden_patches

pytestmark = pytest.mark.gpu

COS_TOL_F32 = 1e-5
PROB_TOL_F32 = 1e-4     # fp32 MFMA vs fp64 oracle on softmax outputs of O(1)-spread logits


def _cos(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


@pytest.fixture(scope="module")
def setup(gpu_device):
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd import weights
    cfg = NestiConfig()
    W = weights.synthetic_weights(cfg)
    g = load_golden_patches([p for p in golden_patch_files() if "ellipsoid100k" in p][0])
    g2 = load_golden_patches([p for p in golden_patch_files() if "ellipsoid20k" in p][0])
    pts = np.concatenate([g["points"][:10], g2["points"][:6]])
    n_eff = np.concatenate([g["n_eff"][:10], g2["n_eff"][:6]])
    return cfg, W, pts, n_eff


@pytest.fixture(scope="module")
def oracle_out(setup):
    from oracle import mups_ref, net_ref
    cfg, W, pts, n_eff = setup
    mups = mups_ref.mups_assemble(pts, n_eff, 3)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    full = net_ref.moe_forward(mups[:6], W, dtype=torch.float64, top1_only=False)      # reference behaviour
    top1 = net_ref.moe_forward(mups, W, dtype=torch.float64, top1_only=True)
    return mups, full, top1


@pytest.fixture(scope="module")
def net_f32(setup, gpu_device):
    from nesti_net_amd.model import NestiNet
    cfg, W, _, _ = setup
    return NestiNet(cfg, W, dtype="f32", device=gpu_device, max_batch=16)


def test_f32_gate_and_routing_match_oracle(setup, oracle_out, net_f32, gpu_device):
    cfg, W, pts, n_eff = setup
    mups_o, full, top1 = oracle_out
    p = torch.as_tensor(pts, device=gpu_device)
    n = torch.as_tensor(n_eff, device=gpu_device)
    mups = net_f32.mups(p, n)
    assert np.abs(mups[..., :60].cpu().numpy() - mups_o).max() < 5e-6
    probs, expert = net_f32.gate(mups)
    torch.cuda.synchronize()
    pe = np.abs(probs.cpu().numpy() - top1["probs"].numpy()).max()
    print("gate prob max abs err (f32):", pe, "experts:", expert.cpu().numpy())
    assert np.array_equal(expert.cpu().numpy(), top1["expert"].numpy())      # bit-exact arg-max
    assert pe < PROB_TOL_F32


def test_f32_all_experts_match_oracle(setup, oracle_out, net_f32, gpu_device):
    """Reference behaviour: every expert on every point -> n_est [E,B,3]."""
    cfg, W, pts, n_eff = setup
    _, full, _ = oracle_out
    mups = net_f32.mups(torch.as_tensor(pts[:6], device=gpu_device), torch.as_tensor(n_eff[:6], device=gpu_device))
    n_est = net_f32.experts(mups, None).cpu().numpy()
    ref = full["n_est"].numpy()
    c = _cos(n_est, ref)
    print("all-experts min cosine (f32):", c.min(), "max abs err", np.abs(n_est - ref).max())
    assert n_est.shape == (7, 6, 3)
    assert np.all(1 - c < COS_TOL_F32)
    assert np.abs(n_est - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())


def test_f32_forward_top1_matches_oracle_and_all_experts(setup, oracle_out, net_f32, gpu_device):
    cfg, W, pts, n_eff = setup
    _, full, top1 = oracle_out
    p = torch.as_tensor(pts, device=gpu_device)
    n = torch.as_tensor(n_eff, device=gpu_device)
    normals, expert, probs = net_f32(p, n)
    torch.cuda.synchronize()
    assert np.array_equal(expert.cpu().numpy(), top1["expert"].numpy())
    c = _cos(normals.cpu().numpy(), top1["normals"].numpy())
    print("top-1 min cosine (f32):", c.min())
    assert np.all(1 - c < COS_TOL_F32)
    # top-1 routing is output-identical to evaluate-all-then-select (test_n_est_w_experts.py:148-152)
    mups = net_f32.mups(p[:6], n[:6])
    n_est = net_f32.experts(mups, None)
    sel = n_est[expert[:6].long(), torch.arange(6, device=gpu_device)]
    assert torch.equal(sel, normals[:6])
    # explicit expert assignment path
    routed = net_f32.experts(net_f32.mups(p, n), expert)
    assert torch.equal(routed, normals)


def test_explicit_routing_to_every_expert(setup, net_f32, gpu_device):
    """Top-1 execution with a caller-supplied assignment that uses every expert (one of them for
    zero points, one for a single point) equals select-after-evaluate-all."""
    cfg, W, pts, n_eff = setup
    p = torch.as_tensor(pts, device=gpu_device)
    n = torch.as_tensor(n_eff, device=gpu_device)
    mups = net_f32.mups(p, n)
    n_est = net_f32.experts(mups, None)
    B = p.shape[0]
    assign = torch.tensor([0, 6, 2, 3, 4, 6, 6, 0, 2, 2, 3, 4, 0, 1, 6, 3][:B], dtype=torch.int32, device=gpu_device)   # no 5
    routed = net_f32.experts(mups, assign)
    sel = n_est[assign.long(), torch.arange(B, device=gpu_device)]
    assert torch.equal(routed, sel)


# Sanity bounds on 16 queries with the data-matched batch-norm statistics (weights.synthetic_weights bn='calibrated': every
# layer's pre-activation is zero-mean / unit-variance over the data, as after training, so rounding noise is measured
# against a signal of realistic size; with arbitrary statistics the network output barely depends on the query and a
# 16-bit run agrees with fp32 to 1e-6 for the wrong reason).  The distribution over 10k queries is asserted in
# tests/test_gpu_fixtures.py::test_production_dtype_parity_on_10k_queries.
@pytest.mark.parametrize("dtype,cos_tol,prob_tol", [("bf16", 1e-1, 4e-1), ("f16", 5e-3, 6e-2)])
def test_16bit_modes_close_to_oracle(setup, oracle_out, gpu_device, dtype, cos_tol, prob_tol):
    from nesti_net_amd.model import NestiNet
    cfg, W, pts, n_eff = setup
    _, full, top1 = oracle_out
    net = NestiNet(cfg, W, dtype=dtype, device=gpu_device, max_batch=16)
    p = torch.as_tensor(pts, device=gpu_device)
    n = torch.as_tensor(n_eff, device=gpu_device)
    normals, expert, probs = net(p, n)
    torch.cuda.synchronize()
    pr = probs.cpu().numpy()
    pe = np.abs(pr - top1["probs"].numpy()).max()
    ex_ref = top1["expert"].numpy()
    agree = expert.cpu().numpy() == ex_ref
    # a flipped arg-max is only acceptable where the oracle's top-2 margin is inside the error bound
    srt = np.sort(top1["probs"].numpy(), axis=1)
    margin = srt[:, -1] - srt[:, -2]
    print(dtype, "prob err", pe, "argmax agree", agree.mean(), "min margin", margin.min())
    assert pe < prob_tol
    assert np.all(agree | (margin < 2 * prob_tol))
    mups = net.mups(p[:6], n[:6])
    n_est = net.experts(mups, None).cpu().numpy()
    c = _cos(n_est, full["n_est"].numpy())
    print(dtype, "all-experts min cosine:", c.min())
    assert np.all(1 - c < cos_tol)


def test_ragged_batches_and_padding_rows(setup, net_f32, gpu_device):
    """Batch sizes that are not multiples of any tile, B=1, and the reference's zero-padded tail
    (n_eff = 0 rows, test_n_est_w_experts.py:134-140): finite outputs, real rows unaffected."""
    cfg, W, pts, n_eff = setup
    p = torch.as_tensor(pts, device=gpu_device)
    n = torch.as_tensor(n_eff, device=gpu_device)
    full, ex_full, pr_full = net_f32(p[:13], n[:13])
    one, ex_one, _ = net_f32(p[4:5], n[4:5])
    assert torch.equal(one[0], full[4]) and ex_one[0] == ex_full[4]
    pad_p = torch.cat([p[:5], torch.zeros_like(p[:3])])
    pad_n = torch.cat([n[:5], torch.zeros_like(n[:3])])
    out, ex, pr = net_f32(pad_p, pad_n)
    assert torch.isfinite(out).all() and torch.isfinite(pr).all()
    assert torch.equal(out[:5], full[:5])


@pytest.mark.parametrize("dtype", ["f32", "f16", "f16x3"])
def test_results_do_not_depend_on_the_point_group_a_query_lands_in(dtype, gpu_device):
    """conv8n_kernel works on groups of 4 points and conv4n_kernel on groups of 16 (an MFMA tile there is one voxel of 16 points,
    the last group of a launch is partly filled, routed experts see arbitrary counts): a query's outputs must be the same bits
    whatever the batch size, its position in the batch and the number of queries routed to its expert -- sizes around the group
    boundaries (1, 15, 16, 17, 33) against a 40-query batch, top-1 routed with a calibrated gate and evaluate-all."""
    from nesti_net_amd import weights
    from nesti_net_amd.calibrate import calibrate_gate
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    cfg = NestiConfig()
    g = load_golden_patches([p for p in golden_patch_files() if "ellipsoid100k" in p][0])
    g2 = load_golden_patches([p for p in golden_patch_files() if "ellipsoid20k" in p][0])
    pts = np.concatenate([g["points"][:24], g2["points"][:16]])
    n_eff = np.concatenate([g["n_eff"][:24], g2["n_eff"][:16]])
    p, n = torch.as_tensor(pts, device=gpu_device), torch.as_tensor(n_eff, device=gpu_device)
    W = calibrate_gate(cfg, weights.synthetic_weights(cfg), p, n, device=gpu_device)
    net = NestiNet(cfg, W, dtype=dtype, device=gpu_device, max_batch=40)
    full = [t.clone() for t in net(p, n)]
    assert len(torch.unique(full[1])) >= 4                                # several experts, each with a ragged count
    all7 = net.experts(net.mups(p, n), None).clone()                       # [E, 40, 3]
    for size in (1, 15, 16, 17, 33):
        for lo in (0, 40 - size):                                          # the same queries at other positions of other batches
            out = net(p[lo:lo + size], n[lo:lo + size])
            for a, b in zip(out, full):
                assert torch.equal(a, b[lo:lo + size]), (dtype, size, lo)
            sub = net.experts(net.mups(p[lo:lo + size], n[lo:lo + size]), None)
            assert torch.equal(sub, all7[:, lo:lo + size]), (dtype, size, lo)


def test_calibrated_gate_routes_to_many_experts_and_matches_oracle(setup, gpu_device):
    """Synthetic gate calibrated to spread its arg-max: the fused top-1 forward (gather by expert,
    per-expert towers with device-side counts, scatter) against the fp64 oracle on 48 queries."""
    from nesti_net_amd.calibrate import calibrate_gate
    from nesti_net_amd.model import NestiNet
    from oracle import mups_ref, net_ref
    cfg, W, _, _ = setup
    g = load_golden_patches([p for p in golden_patch_files() if "ellipsoid100k" in p][0])
    g2 = load_golden_patches([p for p in golden_patch_files() if "ellipsoid20k" in p][0])
    pts = np.concatenate([g["points"][:24], g2["points"][:24]])
    n_eff = np.concatenate([g["n_eff"][:24], g2["n_eff"][:24]])
    p = torch.as_tensor(pts, device=gpu_device)
    n = torch.as_tensor(n_eff, device=gpu_device)
    Wc = calibrate_gate(cfg, W, p, n, device=gpu_device)
    net = NestiNet(cfg, Wc, dtype="f32", device=gpu_device, max_batch=48)
    normals, expert, probs = net(p, n)
    torch.cuda.synchronize()
    ex = expert.cpu().numpy()
    print("routing histogram:", np.bincount(ex, minlength=7))
    assert len(np.unique(ex)) >= 5
    ref = net_ref.moe_forward(mups_ref.mups_assemble(pts, n_eff, 3), Wc, dtype=torch.float64, top1_only=True)
    srt = np.sort(ref["probs"].numpy(), axis=1)
    margin = srt[:, -1] - srt[:, -2]
    agree = ex == ref["expert"].numpy()
    assert np.all(agree | (margin < 1e-4))            # arg-max exact unless the oracle itself is tied to 1e-4
    assert np.abs(probs.cpu().numpy() - ref["probs"].numpy()).max() < 1e-4
    c = _cos(normals.cpu().numpy()[agree], ref["normals"].numpy()[agree])
    assert np.all(1 - c < COS_TOL_F32)


def test_single_scale_model_matches_oracle(gpu_device):
    """BASELINE config 0: ss_norm_est (one radius, one tower, no gate) behind the same boundary."""
    from nesti_net_amd import weights
    from nesti_net_amd.config import ARCH_SINGLE, NestiConfig
    from nesti_net_amd.model import NestiNet
    from oracle import mups_ref, net_ref
    g = load_golden_patches([p for p in golden_patch_files() if "sphere8k" in p][0])
    cfg = NestiConfig(patch_radius=[0.05], n_experts=1, expert_dict={0: [0]}, arch=ARCH_SINGLE)
    W = weights.synthetic_weights(cfg)
    assert W["fc1/weights"].shape == (12288, 1024)
    pts, n_eff = g["points"][:9], g["n_eff"][:9]
    net = NestiNet(cfg, W, dtype="f32", device=gpu_device, max_batch=9)
    normals, expert, probs = net(torch.as_tensor(pts, device=gpu_device), torch.as_tensor(n_eff[:, 0], device=gpu_device))
    torch.cuda.synchronize()
    assert expert is None and probs is None
    ref = net_ref.single_forward(mups_ref.mups_assemble(pts, n_eff, 1), W, dtype=torch.float64).numpy()
    c = _cos(normals.cpu().numpy(), ref)
    print("ss_norm_est min cosine (f32):", c.min())
    assert np.all(1 - c < COS_TOL_F32)
    net16 = NestiNet(cfg, W, dtype="bf16", device=gpu_device, max_batch=9)
    n16, _, _ = net16(torch.as_tensor(pts, device=gpu_device), torch.as_tensor(n_eff, device=gpu_device))
    assert np.all(1 - _cos(n16.cpu().numpy(), ref) < 2e-3)


def test_multi_scale_ablation_matches_oracle(gpu_device):
    """ms_norm_est (N4): S scales concatenated, one tower, 4^3 kernels [3,4]."""
    from nesti_net_amd import weights
    from nesti_net_amd.config import ARCH_MULTI, NestiConfig
    from nesti_net_amd.model import NestiNet
    from oracle import mups_ref, net_ref
    g = load_golden_patches([p for p in golden_patch_files() if "ellipsoid20k" in p][0])
    cfg = NestiConfig(n_experts=1, expert_dict={0: [0, 1, 2]}, arch=ARCH_MULTI)
    W = weights.synthetic_weights(cfg)
    assert W["inception_s2_l_1_conv1/weights"].shape == (1, 1, 1, 60, 128)
    assert W["inception_s2_l_5_conv3/weights"].shape == (4, 4, 4, 512, 256)
    pts, n_eff = g["points"][:7], g["n_eff"][:7]
    net = NestiNet(cfg, W, dtype="f32", device=gpu_device, max_batch=7)
    normals, _, _ = net(torch.as_tensor(pts, device=gpu_device), torch.as_tensor(n_eff, device=gpu_device))
    torch.cuda.synchronize()
    ref = net_ref.multi_forward(mups_ref.mups_assemble(pts, n_eff, 3), W, 3, dtype=torch.float64).numpy()
    c = _cos(normals.cpu().numpy(), ref)
    print("ms_norm_est min cosine (f32):", c.min())
    assert np.all(1 - c < COS_TOL_F32)


def test_switching_model_matches_oracle(gpu_device):
    """ms_sw_n_est (N4): noise_est_net on the large scale thresholds at 0.015 between the 'small' and 'large'
    normal towers (models/ms_sw_n_est.py:75-82).  fc4noise's bias is set so that the synthetic noise estimate
    straddles the threshold and both towers are exercised."""
    from nesti_net_amd import weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    from oracle import mups_ref, net_ref
    g = load_golden_patches([p for p in golden_patch_files() if "ellipsoid20k" in p][0])
    cfg = NestiConfig.for_model("ms_sw_n_est")
    W = weights.synthetic_weights(cfg)
    B = 8
    pts = np.concatenate([g["points"][:B, 0:512], g["points"][:B, 1024:1536]], axis=1)     # radii 0.01 and 0.05
    n_eff = np.stack([g["n_eff"][:B, 0], g["n_eff"][:B, 2]], axis=1)
    mups_o = mups_ref.mups_assemble(pts, n_eff, 2)
    pre = net_ref._ss_tower(torch.as_tensor(mups_o[..., 20:40], dtype=torch.float64), W, "noise", torch.float64, False)[:, 0]
    srt = np.sort(pre.numpy())
    W["fc4noise/biases"] = (W["fc4noise/biases"] + 0.015 - 0.5 * (srt[B // 2 - 1] + srt[B // 2])).astype(np.float32)
    ref = net_ref.switch_forward(mups_o, W, dtype=torch.float64)
    pick_ref = ref["pick"].numpy()
    assert 0 < pick_ref.sum() < B, "calibration should exercise both towers"
    margin = np.abs(ref["noise"].numpy() - 0.015)
    net = NestiNet(cfg, W, dtype="f32", device=gpu_device, max_batch=B)
    p_d, n_d = torch.as_tensor(pts, device=gpu_device), torch.as_tensor(n_eff, device=gpu_device)
    normals, pick, noise = net(p_d, n_d)
    torch.cuda.synchronize()
    assert noise.shape == (B, 1)
    err = np.abs(noise.cpu().numpy()[:, 0] - ref["noise"].numpy())
    print("ms_sw_n_est noise err", err.max(), "margins", margin.min())
    assert err.max() < 2e-5
    safe = margin > 2e-5
    assert np.array_equal(pick.cpu().numpy()[safe], pick_ref[safe])
    c = _cos(normals.cpu().numpy()[safe], ref["normals"].numpy()[safe])
    assert np.all(1 - c < COS_TOL_F32)
    # both towers on every point == [n_est_small, n_est_large] (the tensors tf.where chooses from, :82)
    mups = net.mups(p_d, n_d)
    both = net.experts(mups, None).cpu().numpy()
    assert both.shape == (2, B, 3)
    assert np.all(1 - _cos(both[0], ref["n_small"].numpy()) < COS_TOL_F32)
    assert np.all(1 - _cos(both[1], ref["n_large"].numpy()) < COS_TOL_F32)
    noise2, pick2 = net.gate(mups)
    assert torch.equal(pick2, pick) and torch.equal(noise2, noise)
    # production dtype
    net16 = NestiNet(cfg, W, dtype="bf16", device=gpu_device, max_batch=B)
    n16, p16, z16 = net16(p_d, n_d)
    same = p16.cpu().numpy() == pick_ref
    assert np.all(1 - _cos(n16.cpu().numpy()[same], ref["normals"].numpy()[same]) < 2e-3)
    assert np.all(same | (margin < 5e-3))


def test_3_gaussian_grid_model_matches_oracle(gpu_device):
    """--num_gaussians 3 (the reference's training default, train_n_est_w_experts.py:55-56): 27-Gaussian MuPS,
    conv_net_3g towers (k = 2 / 3 / 1 on a 3^3 volume, max-pool [3,3,3]/2) for the gate and the 7 experts."""
    from nesti_net_amd import weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet, get_model, mups_forward
    from oracle import mups_ref, net_ref
    cfg = NestiConfig(n_gaussians=3, gmm_variance=0.111)
    W = weights.synthetic_weights(cfg)
    g = load_golden_patches([p for p in golden_patch_files() if "ellipsoid20k" in p][0])
    B = 9
    pts, n_eff = g["points"][:B], g["n_eff"][:B]
    p_d, n_d = torch.as_tensor(pts, device=gpu_device), torch.as_tensor(n_eff, device=gpu_device)
    mups_o = mups_ref.mups_assemble(pts, n_eff, 3, grid_n=3, variance=0.111)
    # MuPS: dense [B,3,3,3,60] through the public entry point, and the towers' 4^3-embedded layout
    dense = mups_forward(cfg, p_d, n_d, out_dtype="f32").cpu().numpy()
    assert dense.shape == (B, 3, 3, 3, 60)
    assert np.abs(dense - mups_o).max() < 5e-6
    net = NestiNet(cfg, W, dtype="f32", device=gpu_device, max_batch=B)
    emb = net.mups(p_d, n_d)
    assert emb.shape == (B, 4, 4, 4, 64)
    e = emb.cpu().numpy()
    assert np.array_equal(e[:, :3, :3, :3, :60], dense)
    assert not e[:, 3].any() and not e[:, :, 3].any() and not e[:, :, :, 3].any() and not e[..., 60:].any()
    ref = net_ref.moe_forward(mups_o, W, dtype=torch.float64, top1_only=False)
    probs, expert = net.gate(emb)
    assert np.abs(probs.cpu().numpy() - ref["probs"].numpy()).max() < PROB_TOL_F32
    assert np.array_equal(expert.cpu().numpy(), ref["expert"].numpy())
    n_est = net.experts(emb, None).cpu().numpy()
    assert n_est.shape == (7, B, 3)
    assert np.all(1 - _cos(n_est, ref["n_est"].numpy()) < COS_TOL_F32)
    normals, ex2, pr2 = net(p_d, n_d)
    torch.cuda.synchronize()
    assert np.array_equal(ex2.cpu().numpy(), ref["expert"].numpy())
    assert np.all(1 - _cos(normals.cpu().numpy(), ref["normals"].numpy()) < COS_TOL_F32)
    pr, ne, mu = get_model(net, p_d, n_d)
    assert mu.shape == (B, 3, 3, 3, 60) and pr.shape == (7, B) and ne.shape == (7, B, 3)
    # the reference's own argument list (models/experts_n_est.py:40) + the restored model as the one keyword-only argument
    from nesti_net_amd.model import get_3d_grid_gmm, placeholder_inputs
    gw, gmu, gsg = get_3d_grid_gmm((3, 3, 3), 0.111)
    points_pl, normal_pl, w_pl, mu_pl, sigma_pl, n_eff_pl = placeholder_inputs(B, cfg.num_point, (gw, gmu, gsg), cfg.patch_radius,
                                                                              device=gpu_device)
    points_pl.copy_(p_d)
    n_eff_pl.copy_(n_d)
    pr2_, ne2_, mu2_ = get_model(points_pl, w_pl, mu_pl, sigma_pl, False, cfg.patch_radius, original_n_points=n_eff_pl,
                                 n_experts=cfg.n_experts, expert_dict=cfg.expert_dict, net=net)
    assert torch.equal(pr2_, pr) and torch.equal(ne2_, ne) and torch.equal(mu2_, mu)
    with pytest.raises(ValueError, match="Gaussian grid"):
        get_model(points_pl, w_pl, mu_pl + 0.5, sigma_pl, False, cfg.patch_radius, original_n_points=n_eff_pl, net=net)
    with pytest.raises(ValueError, match="inference only"):
        get_model(points_pl, w_pl, mu_pl, sigma_pl, True, cfg.patch_radius, original_n_points=n_eff_pl, net=net)
    with pytest.raises(ValueError, match="radius"):
        get_model(points_pl, w_pl, mu_pl, sigma_pl, False, [0.1, 0.2, 0.3], original_n_points=n_eff_pl, net=net)
    with pytest.raises(TypeError, match="net="):
        get_model(points_pl, w_pl, mu_pl, sigma_pl, False, cfg.patch_radius, original_n_points=n_eff_pl)
    # production dtypes
    for dt, tol in (("bf16", 2e-3), ("f16", 5e-5)):
        n16 = NestiNet(cfg, W, dtype=dt, device=gpu_device, max_batch=B)
        nn_, ee_, _ = n16(p_d, n_d)
        same = ee_.cpu().numpy() == ref["expert"].numpy()
        assert same.mean() >= 0.75
        assert np.all(1 - _cos(nn_.cpu().numpy()[same], ref["normals"].numpy()[same]) < tol)


def test_other_expert_layouts_match_oracle(gpu_device):
    """The graph builder follows expert_dict / n_experts generically (models/experts_n_est.py:83-103): 4 experts
    {0:[0], 1:[1,2], 2:[2], 3:[0,1,2]} -- a 2-scale expert gets 128/2 = 64 first-block filters."""
    from nesti_net_amd import weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    from oracle import mups_ref, net_ref
    ed = {0: [0], 1: [1, 2], 2: [2], 3: [0, 1, 2]}
    cfg = NestiConfig(n_experts=4, expert_dict=ed)
    W = weights.synthetic_weights(cfg)
    assert W["inception1Expert_1_conv1/weights"].shape == (1, 1, 1, 40, 64)
    assert W["fc4noise/weights"].shape == (128, 4)
    g = load_golden_patches([p for p in golden_patch_files() if "ellipsoid20k" in p][0])
    pts, n_eff = g["points"][:5], g["n_eff"][:5]
    net = NestiNet(cfg, W, dtype="f32", device=gpu_device, max_batch=5)
    mups = net.mups(torch.as_tensor(pts, device=gpu_device), torch.as_tensor(n_eff, device=gpu_device))
    n_est = net.experts(mups, None).cpu().numpy()
    probs, expert = net.gate(mups)
    ref = net_ref.moe_forward(mups_ref.mups_assemble(pts, n_eff, 3), W, expert_dict=ed, dtype=torch.float64)
    assert n_est.shape == (4, 5, 3)
    assert np.all(1 - _cos(n_est, ref["n_est"].numpy()) < COS_TOL_F32)
    assert np.abs(probs.cpu().numpy() - ref["probs"].numpy()).max() < PROB_TOL_F32
    assert np.array_equal(expert.cpu().numpy(), ref["expert"].numpy())


def test_limits_four_scales_eight_experts(gpu_device):
    """The ABI's limits (NESTI_MAX_SCALES = 4, NESTI_MAX_EXPERTS = 8): MuPS with 80 channels (channel stride 128, two
    K-chunks in the first layers), the default expert assignment of models/experts_n_est.py:83-96 (8 // 4 = 2 experts
    per scale, no multi-scale expert), patches straight from the HIP ball query with P = 128."""
    from nesti_net_amd import synth, weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    from nesti_net_amd.provider import CloudPatches
    from oracle import mups_ref, net_ref, patches_ref
    cfg = NestiConfig(patch_radius=[0.01, 0.02, 0.04, 0.06], num_point=128, n_experts=8, expert_dict=None)
    cfg.expert_dict = cfg.default_expert_dict()
    assert cfg.expert_dict == {0: [0], 1: [0], 2: [1], 3: [1], 4: [2], 5: [2], 6: [3], 7: [3]}
    W = weights.synthetic_weights(cfg)
    assert W["inception1gating_conv_conv1/weights"].shape == (1, 1, 1, 80, 128) and W["fc4noise/weights"].shape == (128, 8)
    pts = synth.make_cloud("torus", n=30000, seed=9, noise=0.00125)[0]
    q = np.arange(0, 30000, 5000)
    cp = CloudPatches(pts, cfg, device=gpu_device, pidx=q)
    p_d, n_d = cp.build(0, len(q))
    o_pts, o_neff, _, _ = patches_ref.extract_patches(pts, q, cp.r_abs, cfg.num_point, cp.seed)
    assert np.array_equal(p_d.cpu().numpy().view(np.uint32), o_pts.view(np.uint32)) and np.array_equal(n_d.cpu().numpy(), o_neff)
    net = NestiNet(cfg, W, dtype="f32", device=gpu_device, max_batch=len(q))
    assert net.mups_cstride == 128
    mups = net.mups(p_d, n_d)
    mups_o = mups_ref.mups_assemble(o_pts, o_neff, 4)
    assert np.abs(mups.cpu().numpy()[..., :80] - mups_o).max() < 5e-6 and not mups.cpu().numpy()[..., 80:].any()
    ref = net_ref.moe_forward(mups_o, W, expert_dict=cfg.expert_dict, dtype=torch.float64)
    probs, expert = net.gate(mups)
    assert np.abs(probs.cpu().numpy() - ref["probs"].numpy()).max() < PROB_TOL_F32
    assert np.array_equal(expert.cpu().numpy(), ref["expert"].numpy())
    n_est = net.experts(mups, None).cpu().numpy()
    assert n_est.shape == (8, len(q), 3)
    assert np.all(1 - _cos(n_est, ref["n_est"].numpy()) < COS_TOL_F32)
    normals, _, _ = net(p_d, n_d)
    assert np.all(1 - _cos(normals.cpu().numpy(), ref["normals"].numpy()) < COS_TOL_F32)


def test_empty_batch_is_a_no_op(setup, net_f32, gpu_device):
    cfg, W, pts, n_eff = setup
    p = torch.zeros((0, 1536, 3), dtype=torch.float32, device=gpu_device)
    n = torch.zeros((0, 3), dtype=torch.int32, device=gpu_device)
    normals, expert, probs = net_f32(p, n)
    assert normals.shape == (0, 3) and expert.shape == (0,) and probs.shape == (0, 7)
    assert net_f32.mups(p, n).shape[0] == 0


def test_error_behaviour_at_the_boundary(setup, net_f32, gpu_device):
    """The C-ABI reports misuse through its status code + nesti_last_error (raised as NestiError by the host mirror)
    instead of faulting: workspace too small, missing or mis-shaped variables, wrong model kind, null pointers."""
    import ctypes
    from nesti_net_amd import _lib, weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    cfg, W, pts, n_eff = setup
    lib = _lib.load()
    p = torch.as_tensor(pts[:4], device=gpu_device)
    n = torch.as_tensor(n_eff[:4], device=gpu_device).to(torch.int32)
    small = torch.empty(1 << 20, dtype=torch.uint8, device=gpu_device)
    with pytest.raises(_lib.NestiError, match="workspace too small"):
        net_f32.forward(p, n, ws=small)
    # the call after a failed one works and the workspace was not touched beyond its end
    normals, expert, probs = net_f32(p, n)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(normals).all())
    bad = dict(W)
    del bad["inception2gating_conv_conv3/weights"]
    with pytest.raises(_lib.NestiError, match="inception2gating_conv_conv3/weights"):
        NestiNet(cfg, bad, dtype="bf16", device=gpu_device, max_batch=4)
    bad = dict(W)
    bad["fc4noise/biases"] = np.zeros(6, np.float32)
    with pytest.raises(_lib.NestiError, match="fc4noise/biases"):
        NestiNet(cfg, bad, dtype="bf16", device=gpu_device, max_batch=4)
    ss = NestiNet(NestiConfig.for_model("ss_norm_est"), weights.synthetic_weights(NestiConfig.for_model("ss_norm_est")),
                  dtype="bf16", device=gpu_device, max_batch=4)
    with pytest.raises(_lib.NestiError, match="no gating net"):
        ss.gate(torch.zeros((4, 8, 8, 8, 64), dtype=torch.bfloat16, device=gpu_device))
    assert lib.nesti_forward(net_f32._handle, None, None, 4, None, 0, None, None, None, None) != 0
    assert b"null argument" in lib.nesti_last_error()
