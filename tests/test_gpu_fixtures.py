"""The whole hot path in the exact-fp32 MFMA mode (and the f16x3 pair mode) against the CPU oracle on every golden fixture's cloud
(no-noise / PCPNet noise levels / gradient and striped density / small P / single scale), on >= 256 queries drawn
ACROSS each cloud (strided) plus the fixture's own reference-captured queries, with a calibrated gate so that the
routed top-1 path exercises every expert -- and the production dtypes against the fp32 mode on >= 10k queries.

Chain of custody: oracle/patches_ref is pinned to the reference's own PointcloudPatchDataset by the golden rows
(tests/test_oracle_patches.py), oracle/mups_ref to the reference's numpy 3DmFV (tests/test_oracle_mups.py); here
HIP patches == oracle patches bit for bit, HIP MuPS == oracle MuPS to 1e-5 (5e-6 in tests/test_gpu_mups.py), HIP fp32 net == oracle fp64 net (arg-max
exact, probabilities 1e-4, normals 1e-5 cosine); f16 / bf16 are then characterised against the fp32 mode
(nesti_net_amd/parity.py), which is what bench.py prints for the timed run."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_patch_files, load_golden_patches

pytestmark = pytest.mark.gpu

N_STRIDED = 256
FIXTURES = [os.path.basename(p)[len("patches_"):-len(".npz")] for p in golden_patch_files()]


def _cos(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def _fixture_config(g):
    from nesti_net_amd.config import ARCH_SINGLE, NestiConfig
    radii = [float(r) for r in g["radii"]]
    if len(radii) == 1:                                    # BASELINE config 0: ss_norm_est
        return NestiConfig(patch_radius=radii, num_point=g["P"], n_experts=1, expert_dict={0: [0]}, arch=ARCH_SINGLE)
    if len(radii) == 3:
        return NestiConfig(patch_radius=radii, num_point=g["P"])
    cfg = NestiConfig(patch_radius=radii, num_point=g["P"], n_experts=2 * len(radii), expert_dict=None)
    cfg.expert_dict = cfg.default_expert_dict()            # models/experts_n_est.py:83-96
    return cfg


def _oracle_moe(mups, W, cfg):
    """oracle.net_ref.moe_forward in chunks of 16 queries spread over host threads (net_ref.over_chunks); fp64."""
    from oracle import net_ref
    outs = net_ref.over_chunks(lambda sl: net_ref.moe_forward(mups[sl], W, expert_dict=cfg.expert_dict, dtype=torch.float64, top1_only=True),
                               len(mups))
    return {k: torch.cat([o[k] for o in outs]).numpy() for k in ("probs", "expert", "normals")}


@pytest.mark.parametrize("name", FIXTURES)
def test_f32_path_matches_oracle_across_fixture_cloud(name, gpu_device):
    from nesti_net_amd import weights
    from nesti_net_amd.calibrate import calibrate_gate
    from nesti_net_amd.config import ARCH_SINGLE
    from nesti_net_amd.model import NestiNet
    from nesti_net_amd.provider import CloudPatches
    from oracle import mups_ref, net_ref, patches_ref
    g = load_golden_patches([p for p in golden_patch_files() if name in p][0])
    cfg = _fixture_config(g)
    pts, N = g["pts"], len(g["pts"])
    strided = np.arange(N // (2 * N_STRIDED), N, N // N_STRIDED)[:N_STRIDED]
    q = np.unique(np.concatenate([g["queries"].astype(np.int64), strided]))
    assert len(q) >= N_STRIDED
    S = cfg.n_scales
    # ---- patches: HIP == oracle == (on the fixture's rows, wherever the ball holds <= P points) the reference ----------
    cp = CloudPatches(pts, cfg, device=gpu_device, seed=g["seed"], pidx=q)
    assert np.allclose(cp.r_abs, g["r_abs"], rtol=0, atol=0)
    p_d, n_d = cp.build(0, len(q))
    o_pts, o_neff, _, _ = patches_ref.extract_patches(pts, q, cp.r_abs, cfg.num_point, cp.seed)
    assert np.array_equal(n_d.cpu().numpy(), o_neff)
    assert np.array_equal(p_d.cpu().numpy().view(np.uint32), o_pts.view(np.uint32))
    # ---- MuPS ------------------------------------------------------------------------------------------------------
    W = weights.synthetic_weights(cfg)
    if cfg.arch != ARCH_SINGLE:
        W = calibrate_gate(cfg, W, p_d, n_d, device=gpu_device)          # spread the arg-max over the experts
    net = NestiNet(cfg, W, dtype="f32", device=gpu_device, max_batch=len(q))
    mups_o = mups_ref.mups_assemble(o_pts, o_neff, S)
    mups = net.mups(p_d, n_d).cpu().numpy()
    err = np.abs(mups[..., :20 * S] - mups_o).max()
    print(name, "queries", len(q), "n_eff min/mean", o_neff.min(0), o_neff.mean(0).round(1), "MuPS max abs err", err)
    assert err < 1e-5 and not mups[..., 20 * S:].any()     # 1.2e-6 on dense clouds; patches of 1-3 points (the noisy sets) reach 5.5e-6
    # ---- network ---------------------------------------------------------------------------------------------------
    normals, expert, probs = net(p_d, n_d)
    torch.cuda.synchronize()
    if cfg.arch == ARCH_SINGLE:
        ref = torch.cat(net_ref.over_chunks(lambda sl: net_ref.single_forward(mups_o[sl], W, dtype=torch.float64), len(q))).numpy()
        c = _cos(normals.cpu().numpy(), ref)
        n3, _, _ = NestiNet(cfg, W, dtype="f16x3", device=gpu_device, max_batch=len(q))(p_d, n_d)
        c3 = _cos(n3.cpu().numpy(), ref)
        print(name, "ss_norm_est 1-cos max", (1 - c).max(), "f16x3", (1 - c3).max())
        assert np.all(1 - c < 1e-5) and np.all(1 - c3 < 1e-5)
        return
    ref = _oracle_moe(mups_o, W, cfg)
    ex = expert.cpu().numpy()
    srt = np.sort(ref["probs"], axis=1)
    margin = srt[:, -1] - srt[:, -2]
    agree = ex == ref["expert"]
    perr = np.abs(probs.cpu().numpy() - ref["probs"]).max()
    c = _cos(normals.cpu().numpy()[agree], ref["normals"][agree])
    print(name, "routing", np.bincount(ex, minlength=cfg.n_experts), "prob err", perr, "flips", int((~agree).sum()),
          "min margin", margin.min(), "1-cos max", (1 - c).max())
    assert len(np.unique(ex)) >= min(5, cfg.n_experts)
    # the bound parity.TIE_MARGIN is derived from (2 x it): the f32 mode's probabilities against the fp64 oracle's on
    # every query of every fixture cloud; the calibrated gate's last layer amplifies logit differences
    from nesti_net_amd import parity
    print(name, "F32_PROB_ERR on this fixture %.4g (bound %.4g, TIE_MARGIN %.4g)" % (perr, parity.F32_PROB_ERR_BOUND, parity.TIE_MARGIN))
    assert perr <= parity.F32_PROB_ERR_BOUND
    # ONE tie rule for suite, smoke and bench (parity.TIE_MARGIN); differences the old hand-picked 2e-5 would not excuse are printed
    print(name, "f32 flips with an oracle gap in [%.0e, TIE_MARGIN): %d" % (parity.TIE_MARGIN_HAND, int((~agree & (margin >= parity.TIE_MARGIN_HAND)).sum())))
    assert np.all(agree | (margin < parity.TIE_MARGIN))   # arg-max exact unless the fp64 oracle's own top-2 are tied at fp32 level
    assert np.all(1 - c < 1e-5)
    # ---- the same rows in the f16x3 pair mode (the north-star mode of the bench) against the same oracle results ----
    del net
    n3, e3, p3 = NestiNet(cfg, W, dtype="f16x3", device=gpu_device, max_batch=len(q))(p_d, n_d)
    torch.cuda.synchronize()
    agree3 = e3.cpu().numpy() == ref["expert"]
    perr3 = np.abs(p3.cpu().numpy() - ref["probs"]).max()
    c3 = _cos(n3.cpu().numpy()[agree3], ref["normals"][agree3])
    print(name, "f16x3: prob err", perr3, "flips", int((~agree3).sum()), "1-cos max", (1 - c3).max())
    assert perr3 < 1e-4
    print(name, "f16x3 flips with an oracle gap in [%.0e, TIE_MARGIN): %d" % (parity.TIE_MARGIN_HAND, int((~agree3 & (margin >= parity.TIE_MARGIN_HAND)).sum())))
    assert np.all(agree3 | (margin < parity.TIE_MARGIN))
    assert np.all(1 - c3 < 1e-5)
    # ---- ... and in f16x3c, the headline mode: f16x3 behind the two-stage gate, its margin calibrated on this fixture's own
    # queries (>= 256, as bench.py does on the timed cloud), against the SAME fp64 oracle results ------------------------------
    from nesti_net_amd.calibrate import calibrate_gate_margin
    net_c = NestiNet(cfg, W, dtype="f16x3c", device=gpu_device, max_batch=len(q))
    tau = calibrate_gate_margin(net_c, p_d, n_d)
    nc, ec, pc = net_c(p_d, n_d)
    st = net_c.cascade_stats()
    agree_c = ec.cpu().numpy() == ref["expert"]
    perr_c = np.abs(pc.cpu().numpy() - ref["probs"])
    cc = _cos(nc.cpu().numpy()[agree_c], ref["normals"][agree_c])
    twice = (pc == p3).all(dim=1).cpu().numpy()              # rows decided by the f16x3 gate carry its probabilities
    print(name, "f16x3c: tau %.4g" % tau, st, "flips", int((~agree_c).sum()), "prob err (rechecked rows / all)",
          perr_c[twice].max() if twice.any() else 0.0, perr_c.max(), "1-cos max", (1 - cc).max())
    assert np.isfinite(tau) and 0 < st["rechecked"] < len(q) and st["queries"] == len(q)
    assert np.all(agree_c | (margin < parity.TIE_MARGIN))    # arg-max exact against the oracle, like f32 and f16x3 above
    assert np.array_equal(ec.cpu().numpy(), e3.cpu().numpy())   # ... and identical to f16x3's on every query
    assert torch.equal(nc, n3)                               # the experts always run in f16x3
    assert np.all(1 - cc < 1e-5)
    assert twice.sum() >= st["rechecked"] and (not twice.any() or perr_c[twice].max() < 1e-4)
    assert perr_c.max() < 0.05                               # unrechecked rows carry the f16 gate's probabilities
    assert st["max_margin_err"] * 1.5 <= st["tau_eff"] * (1 + 1e-6)
    # ---- ... and in f16x8c, the headline mode since round 6 (8^3 grid): the experts' tap layers at 8^3 with their cross terms through
    # FP8, the conditioning guard calibrated on the fixture's own queries, against the SAME fp64 oracle results ---------------------
    if cfg.n_gaussians == 8:
        from nesti_net_amd.calibrate import calibrate_x8_guard
        del net_c
        net_8 = NestiNet(cfg, W, dtype="f16x8c", device=gpu_device, max_batch=len(q))
        calibrate_gate_margin(net_8, p_d, n_d)
        thr8 = calibrate_x8_guard(net_8, p_d, n_d)
        n8, e8, _ = net_8(p_d, n_d)
        g8 = net_8.x8_guard_stats()
        agree_8 = e8.cpu().numpy() == ref["expert"]
        c8 = _cos(n8.cpu().numpy()[agree_8], ref["normals"][agree_8])
        c83 = _cos(n8.cpu().numpy(), n3.cpu().numpy())
        print(name, "f16x8c: guard thr %.4g" % thr8, g8, "1-cos max vs oracle %.3g, vs f16x3 %.3g" % ((1 - c8).max(), (1 - c83).max()))
        assert np.array_equal(e8.cpu().numpy(), e3.cpu().numpy())      # the gate is untouched
        assert np.all(1 - c8 < 1e-5) and np.all(1 - c83 <= 2.5e-6)
        assert g8["dropped"] == 0 and g8["queries"] == len(q)


@pytest.fixture(scope="module")
def big_case(gpu_device):
    """10 240 queries strided over the 100k-point ellipsoid, calibrated gate, fp32-mode reference outputs."""
    from nesti_net_amd import synth, weights
    from nesti_net_amd.calibrate import calibrate_gate
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.pipeline import NormalEstimator
    from nesti_net_amd.provider import CloudPatches
    cfg = NestiConfig()
    pts = synth.make_cloud("ellipsoid", n=100000, seed=1234)[0]
    q = np.arange(3, 100000, 100000 // 10240)[:10240]
    cp = CloudPatches(pts, cfg, device=gpu_device)
    sp, sn = cp.build(0, 512)
    W = calibrate_gate(cfg, weights.synthetic_weights(cfg), sp, sn, device=gpu_device)
    del cp, sp, sn
    est = NormalEstimator(cfg, W, dtype="f32", device=gpu_device, batch=2048)
    ref = est.estimate(pts, pidx=q)
    del est
    torch.cuda.empty_cache()
    return cfg, W, pts, q, ref


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_production_dtype_parity_on_10k_queries(big_case, gpu_device, dtype):
    """The numbers bench.py prints under "parity", asserted on 10 240 queries with the data-matched synthetic weights.
    Neither plain 16-bit mode meets the north star's tolerance (bit-exact arg-max, 1e-5 cosine) -- the pair modes (next
    test) and the exact-fp32 MFMA mode do, and the latter is what the oracle tests hold to it.  What is asserted here is each dtype's measured
    distribution with headroom: f16 (the default) keeps > 98.5 % of the arg-maxes with a median 1 - cos of a few 1e-6,
    bf16 (3 % faster) > 90 % with a median of a few 1e-4."""
    from nesti_net_amd import parity
    from nesti_net_amd.pipeline import NormalEstimator
    cfg, W, pts, q, ref = big_case
    est = NormalEstimator(cfg, W, dtype=dtype, device=gpu_device, batch=4096)
    out = est.estimate(pts, pidx=q)
    rep = parity.compare(out, ref)
    print(dtype, rep)
    assert rep["queries"] == 10240
    assert len(np.unique(ref[1])) == 7
    omc = rep["one_minus_cos"]
    if dtype == "f16":
        assert rep["argmax_match_rate"] >= 0.985
        assert rep["prob_abs_err_max"] < 0.06
        assert omc["p50"] <= 2e-5 and omc["p99"] <= 1e-3
    else:
        assert rep["argmax_match_rate"] >= 0.90
        assert rep["prob_abs_err_max"] < 0.4
        assert omc["p50"] <= 2e-3 and omc["p99"] <= 6e-2
    # a flip outside the near-tie margin exists in both modes; it is counted, not hidden
    assert rep["argmax_flips"] == rep["argmax_ties"] + rep["flips_outside_margin"] and not rep["meets_north_star"]


@pytest.mark.parametrize("mode", ["f16x3c", "f16x3", "bf16x3"])
def test_pair_modes_meet_the_north_star_on_10k_queries(big_case, gpu_device, mode):
    """dtype 'f16x3' / 'bf16x3' (activations and weights as 16-bit hi + lo pairs, three MFMA products per multiply) and
    'f16x3c' (f16x3 behind the two-stage gate, the bench's headline mode) against the exact-fp32 mode on the same 10 240
    queries: normals within 1e-5 cosine (test_n_est_w_experts.py's outputs to the north star's tolerance) and no arg-max
    difference outside the fp32 reference's own tie margin (parity.TIE_MARGIN = 2 x the measured f32-vs-fp64 probability error).  The f16 pair modes keep two orders of
    magnitude of headroom on the normals; bf16x3 (2^-17 operands) is at the edge of the tolerance (scripts/pair_mode_sweep.py)
    and is only held to its measured distribution."""
    from nesti_net_amd import parity
    from nesti_net_amd.calibrate import calibrate_gate_margin
    from nesti_net_amd.pipeline import NormalEstimator
    from nesti_net_amd.provider import CloudPatches
    cfg, W, pts, q, ref = big_case
    est = NormalEstimator(cfg, W, dtype=mode, device=gpu_device, batch=4096)
    if mode == "f16x3c":
        sp, sn = CloudPatches(pts, cfg, device=gpu_device).build(0, 1024)
        tau = calibrate_gate_margin(est.net, sp, sn)
        del sp, sn
    out = est.estimate(pts, pidx=q)
    rep = parity.compare(out, ref)
    print(mode, rep)
    assert rep["queries"] == 10240
    assert rep["argmax_match_rate"] >= 0.999
    if mode == "bf16x3":
        assert rep["argmax_flips"] <= 4 and rep["flip_margin_max"] < 5e-4 and rep["one_minus_cos"]["max"] <= 1e-5
        return
    assert rep["flips_outside_margin"] == 0
    assert rep["one_minus_cos"]["max"] <= 1e-7
    assert rep["meets_north_star"]
    if mode == "f16x3c":
        st = est.net.cascade_stats()
        print("cascade", st)
        assert st["queries"] == 10240 and st["rechecked"] < 4000 and st["max_margin_err"] < 0.8 * tau


def test_fp32_mode_is_batching_invariant_and_self_consistent(big_case, gpu_device):
    """The reference side of every parity figure: the exact-fp32 mode gives the same bits whatever the batch size."""
    from nesti_net_amd import parity
    from nesti_net_amd.pipeline import NormalEstimator
    cfg, W, pts, q, ref = big_case
    est = NormalEstimator(cfg, W, dtype="f32", device=gpu_device, batch=1000)
    out = est.estimate(pts, pidx=q[:3000])
    for x, y in zip(out, ref):
        assert np.array_equal(x, y[:3000])
    rep = parity.compare(out, [r[:3000] for r in ref])
    assert rep["meets_north_star"] and rep["argmax_flips"] == 0 and rep["one_minus_cos"]["max"] < 1e-12
