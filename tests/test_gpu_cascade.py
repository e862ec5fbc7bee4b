"""dtype 'f16x3c' (NESTI_F16X3C, include/nesti_hip.h): f16x3 with the two-stage gate -- the gating net
(models/experts_n_est.py:155-179) runs in plain f16 as a filter and only the queries whose f16 top-2 logit margin is below
the gate margin tau are decided by the f16x3 gating net; the experts always run in f16x3.

What must hold, whatever tau is: the normals of a query are the f16x3 normals of the expert it was routed to; with
tau = inf every output equals the f16x3 mode bit for bit; with tau = 0 the arg-max is the plain f16 gate's; with the
calibrated tau the arg-max equals the f16x3 arg-max on every query and the f16 gate's measured error on a logit difference
(nesti_model_cascade_stats) stays below 0.8 tau."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B = 6000      # > 4096: the recheck runs in four rounds of 1536 rows (model.hip: cascade_cap)


@pytest.fixture(scope="module")
def case(gpu_device):
    from nesti_net_amd import synth, weights
    from nesti_net_amd.calibrate import calibrate_gate
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    from nesti_net_amd.provider import CloudPatches
    cfg = NestiConfig()
    pts = synth.make_cloud("ellipsoid", n=100000, seed=1234)[0]
    q = np.arange(5, 100000, 100000 // B)[:B]
    cp = CloudPatches(pts, cfg, device=gpu_device, pidx=q)
    points, n_eff = cp.build(0, B)
    W = calibrate_gate(cfg, weights.synthetic_weights(cfg), points[:512], n_eff[:512], device=gpu_device)
    x3 = NestiNet(cfg, W, dtype="f16x3", device=gpu_device, max_batch=B)
    ref = [t.clone() for t in x3(points, n_eff)]
    n_all = x3.experts(x3.mups(points, n_eff), None).clone()          # [E, B, 3]: every expert's f16x3 normal of every query
    f16 = NestiNet(cfg, W, dtype="f16", device=gpu_device, max_batch=B)
    p16, e16 = [t.clone() for t in f16.gate(f16.mups(points, n_eff))]
    # the FILTER pass's own decisions (tau = 0: nothing is rechecked): plain-f16 activations and tap layers, 1x1x1 / FC layers on
    # the exact pair-packed weights -- close to, but since round 5 not identical with, the plain f16 model's gate above
    fl = NestiNet(cfg, W, dtype="f16x3c", device=gpu_device, max_batch=B)
    fl.set_gate_margin(0.0)
    pfl, efl = [t.clone() for t in fl.gate(fl.mups(points, n_eff))]
    torch.cuda.synchronize()
    del x3, f16, fl
    torch.cuda.empty_cache()
    return cfg, W, pts, q, points, n_eff, ref, n_all, p16, e16, pfl, efl


def test_tau_infinite_equals_f16x3_bitwise_and_counts_every_query(case, gpu_device):
    from nesti_net_amd.model import NestiNet
    cfg, W, pts, q, points, n_eff, ref, n_all, p16, e16, pfl, efl = case
    net = NestiNet(cfg, W, dtype="f16x3c", device=gpu_device, max_batch=B)
    net.set_gate_margin(1e30)
    out = net(points, n_eff)
    st = net.cascade_stats(reset=True)
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    assert st["queries"] == B and st["rechecked"] == B
    assert st["changed"] == int((efl != ref[1]).sum().item())          # what the filter pass alone would have got wrong
    assert 0 < st["max_margin_err"] < 1.0
    # .gate() / .experts() (the reference-shaped pieces) go through the same two stages
    mups = net.mups(points, n_eff)
    probs, expert = net.gate(mups)
    assert torch.equal(probs, ref[2]) and torch.equal(expert, ref[1])
    assert torch.equal(net.experts(mups, expert), ref[0])
    assert net.cascade_stats()["queries"] == B                          # the reset above took effect, this call counted again


def test_tau_zero_keeps_the_f16_gate_and_routes_f16x3_experts(case, gpu_device):
    from nesti_net_amd.model import NestiNet
    cfg, W, pts, q, points, n_eff, ref, n_all, p16, e16, pfl, efl = case
    net = NestiNet(cfg, W, dtype="f16x3c", device=gpu_device, max_batch=B)
    net.set_gate_margin(0.0)
    normals, expert, probs = net(points, n_eff)
    st = net.cascade_stats()
    assert st["rechecked"] == 0 and st["changed"] == 0 and st["max_margin_err"] == 0.0
    # the filter pass alone decided: plain-f16 activations (hi plane of the pair MuPS), tap layers in plain f16, and -- since round
    # 5 -- the 1x1x1 / FC layers multiplied by their EXACT (pair-packed) weights.  It is therefore no longer the plain f16 model's
    # gate bit for bit, but closer to the f16x3 gate than that one: fewer arg-max differences, smaller probability error.
    flips_filter = int((expert != ref[1]).sum().item())
    flips_f16 = int((e16 != ref[1]).sum().item())
    perr_filter = float((probs - ref[2]).abs().max().item())
    perr_f16 = float((p16 - ref[2]).abs().max().item())
    print("filter alone: %d arg-max differences vs f16x3 (plain f16 gate: %d), prob err %.4g (plain f16: %.4g)"
          % (flips_filter, flips_f16, perr_filter, perr_f16))
    assert flips_filter <= flips_f16 and perr_filter <= perr_f16 and perr_filter < 0.03
    assert not torch.equal(probs, p16)
    assert torch.equal(expert, efl) and torch.equal(probs, pfl)         # forward() and gate() run the same filter
    pick = n_all[expert.long(), torch.arange(B, device=gpu_device)]
    assert torch.equal(normals, pick)                                   # f16x3 normals of whatever expert was chosen


def test_calibrated_margin_reproduces_the_f16x3_decisions(case, gpu_device):
    from nesti_net_amd.calibrate import calibrate_gate_margin
    from nesti_net_amd.model import NestiNet
    cfg, W, pts, q, points, n_eff, ref, n_all, p16, e16, pfl, efl = case
    net = NestiNet(cfg, W, dtype="f16x3c", device=gpu_device, max_batch=B)
    assert calibrate_gate_margin(net, points[:100], n_eff[:100]) == float("inf")   # too few queries to calibrate on: no filtering
    assert net.cascade_stats()["tau"] > 1e29
    tau = calibrate_gate_margin(net, points[:1024], n_eff[:1024])
    assert 0 < tau < 2.0
    normals, expert, probs = net(points, n_eff)
    st = net.cascade_stats()
    print("tau", tau, st, "f16 gate flips", int((e16 != ref[1]).sum().item()))
    assert st["queries"] == B and 0 < st["rechecked"] < B // 2          # a filter, not a second full pass
    assert tau >= 6.9 * st["sigma"]                                     # what tau rests on, re-measured on this batch
    # the margin protects itself (NESTI_GATE_WIDEN): whatever this batch measured, every unrechecked row kept a margin of at
    # least 1.5 x the largest error, either because tau already was that large or because a widening round re-decided the band
    assert st["tau_eff"] == pytest.approx(max(tau, 1.5 * st["max_margin_err"]), rel=1e-6)
    assert (st["widen_events"] > 0) == (st["widened"] > 0)
    if st["max_margin_err"] <= tau / 1.5:
        assert st["widened"] == 0 and st["tau_eff"] == pytest.approx(tau)
    assert torch.equal(expert, ref[1]) and torch.equal(normals, ref[0])
    # probabilities: f16x3's on the rechecked rows, the filter pass's elsewhere (within its error of f16x3's)
    same = (probs == ref[2]).all(dim=1)
    assert int(same.sum().item()) >= st["rechecked"]
    assert torch.equal(probs[~same], pfl[~same])


def test_margin_widens_itself_when_the_measured_error_approaches_it(case, gpu_device):
    """A margin that is too small for the data does not just show up in the statistics: the first call measures the f16
    gate's error on the rows it decides twice and re-decides, in the same call, the rows whose margin lies between tau and
    1.5 x that error (widened > 0) -- in up to NESTI_GATE_WIDEN_PASSES passes, so that an error first seen INSIDE a widening
    pass is covered by the next pass of the same call; the next call starts from the raised threshold
    and needs no widening.  When the passes converged (the last one found nothing), every row that still keeps the f16
    arg-max has a margin of at least 1.5 x the largest error measured by the call."""
    from nesti_net_amd.model import NestiNet
    cfg, W, pts, q, points, n_eff, ref, n_all, p16, e16, pfl, efl = case
    net = NestiNet(cfg, W, dtype="f16x3c", device=gpu_device, max_batch=B)
    tau = 0.02                                                          # far below the f16 gate's error on this data (~0.1)
    net.set_gate_margin(tau)
    net.cascade_stats(reset=True)
    normals, expert, probs = net(points, n_eff)
    st1 = net.cascade_stats()
    print("first call", st1)
    assert st1["tau"] == pytest.approx(tau) and st1["max_margin_err"] > tau / 1.5
    from nesti_net_amd import _lib
    assert 1 <= st1["widen_events"] <= _lib.GATE_WIDEN_PASSES and 0 < st1["widened"] < st1["rechecked"] <= B
    assert st1["tau_eff"] == pytest.approx(1.5 * st1["max_margin_err"], rel=1e-6)
    # every row with an f16 margin below 1.5 x the error known when the widening round ran was decided by the f16x3 gate:
    # its outputs are f16x3's bit for bit; the others keep the f16 gate's (and may differ from f16x3 only through its error)
    l16 = torch.log(pfl.double())                                       # the filter pass's own logits (up to a shift)
    srt = torch.sort(l16, dim=1, descending=True).values
    margin16 = (srt[:, 0] - srt[:, 1]).cpu().numpy()                    # = the f16 logit margin (softmax is shift-invariant)
    decided_twice = (probs == ref[2]).all(dim=1).cpu().numpy()
    assert decided_twice.sum() >= st1["rechecked"]
    assert np.all(margin16[~decided_twice] >= tau - 1e-4)
    if st1["widen_events"] < _lib.GATE_WIDEN_PASSES:                    # converged: the last pass measured nothing new
        assert np.all(margin16[~decided_twice] >= 1.5 * st1["max_margin_err"] - 1e-4), "a kept row sits inside 1.5 x the measured error"
    assert torch.equal(expert[torch.as_tensor(decided_twice)], ref[1][torch.as_tensor(decided_twice)])
    flips = int((expert != ref[1]).sum().item())
    assert flips <= int((e16 != ref[1]).sum().item())
    # second call: the threshold already is tau_eff -- the same rows (and more) go through the first list, nothing is widened
    net.cascade_stats(reset=False)
    normals2, expert2, probs2 = net(points, n_eff)
    st2 = net.cascade_stats()
    print("second call", st2)
    if st2["max_margin_err"] == st1["max_margin_err"]:
        assert st2["widen_events"] == st1["widen_events"] and st2["widened"] == st1["widened"]
    assert st2["rechecked"] - st1["rechecked"] >= st1["rechecked"]
    assert int((expert2 != ref[1]).sum().item()) <= flips
    # resetting the counters forgets the measured error: the threshold is tau again
    net.cascade_stats(reset=True)
    assert net.cascade_stats()["tau_eff"] == pytest.approx(tau)


def test_four_scale_filter_pass_reads_both_channel_groups(gpu_device):
    """n_scales = 4: the MuPS tensor has 80 channels = TWO 64-channel groups, which the pair layout stores as
    [hi0 | lo0 | hi1 | lo1]; the filter pass must read hi0 and hi1 (ConvParams::in_chunk_bytes; its first layer runs the X2
    loop on 32-channel chunks of the hi planes), not hi0 and lo0.  With tau = 0 the filter pass IS the whole gate: its
    probabilities must sit within the filter's error of the f16x3 gate's -- closer than the plain-f16 model's gate, whose weights
    are rounded -- while a wrong plane would feed 20 channels of rounding residue and land far away."""
    from nesti_net_amd import weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    cfg = NestiConfig(patch_radius=[0.01, 0.02, 0.04, 0.06], num_point=128, n_experts=8, expert_dict=None)
    cfg.expert_dict = cfg.default_expert_dict()
    W = weights.synthetic_weights(cfg)
    rng = np.random.RandomState(11)
    Bq = 37
    pts = ((rng.rand(Bq, 4 * 128, 3) - 0.5) * 1.6).astype(np.float32)
    n_eff = rng.randint(20, 129, size=(Bq, 4)).astype(np.int32)
    for b in range(Bq):
        for s_ in range(4):
            pts[b, 128 * s_ + n_eff[b, s_]:128 * (s_ + 1)] = 0
    p, n = torch.as_tensor(pts, device=gpu_device), torch.as_tensor(n_eff, device=gpu_device)
    f16 = NestiNet(cfg, W, dtype="f16", device=gpu_device, max_batch=Bq)
    p16, e16 = f16.gate(f16.mups(p, n))
    net = NestiNet(cfg, W, dtype="f16x3c", device=gpu_device, max_batch=Bq)
    assert net.mups_cstride == 2 * 128
    net.set_gate_margin(0.0)
    pc, ec = net.gate(net.mups(p, n))
    x3g = NestiNet(cfg, W, dtype="f16x3", device=gpu_device, max_batch=Bq)
    p3, e3 = x3g.gate(x3g.mups(p, n))
    err_c, err_16 = float((pc - p3).abs().max().item()), float((p16 - p3).abs().max().item())
    print("4-scale filter: prob err vs f16x3 %.4g (plain f16 gate %.4g)" % (err_c, err_16))
    assert err_c < 0.01                                  # a wrong plane would put the probabilities tenths away
    del x3g
    # and scale 4 really reaches the gate: zeroing its patches changes the f16 probabilities
    pts0 = pts.copy()
    pts0[:, 3 * 128:] = 0
    p0, _ = f16.gate(f16.mups(torch.as_tensor(pts0, device=gpu_device), n))
    assert not torch.equal(p0, p16)
    # with tau = inf the mode is f16x3
    x3 = NestiNet(cfg, W, dtype="f16x3", device=gpu_device, max_batch=Bq)
    net.set_gate_margin(1e30)
    for a, b in zip(net(p, n), x3(p, n)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", ["f16x3c", "f16x8c"])
def test_product_path_batches_graph_and_stream_modes_agree(case, gpu_device, dtype):
    """nesti_estimate_normals with a ragged tail, the per-batch Python loop, a captured hipGraph and run_many: the same
    bits (each query's result depends on neither its batch nor its position in the flag list).  f16x8c: the same with the FP8
    cross-term experts and their conditioning guard (default threshold; inside a captured graph the guard runs on the captured
    stream instead of the model's auxiliary one)."""
    from nesti_net_amd.pipeline import NormalEstimator
    cfg, W, pts, q, points, n_eff, ref, n_all, p16, e16, pfl, efl = case
    sub = q[:2500]
    outs = []
    for kw in ({"batch": 2500}, {"batch": 1000}, {"batch": 1024, "use_graph": True}, {"batch": 900, "n_streams": 2}):
        est = NormalEstimator(cfg, W, dtype=dtype, device=gpu_device, gate_margin=0.3, **kw)
        outs.append(est.estimate(pts, pidx=sub))
        if kw == {"batch": 1000}:
            cloud = est.prepare(pts, pidx=sub)
            many = est.run_many([(cloud, 0, 700), (cloud, 700, 1800)])
            torch.cuda.synchronize()
            for k in range(3):
                assert np.array_equal(np.concatenate([many[0][k].cpu().numpy(), many[1][k].cpu().numpy()]), outs[0][k])
        del est
        torch.cuda.empty_cache()
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert np.array_equal(a, b)
    assert np.array_equal(outs[0][1], ref[1][:2500].cpu().numpy())


def test_cascade_is_for_the_gated_model_only(gpu_device):
    from nesti_net_amd import _lib, weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    cfg = NestiConfig.for_model("ss_norm_est")
    with pytest.raises(_lib.NestiError, match="NESTI_F16X3C"):
        NestiNet(cfg, weights.synthetic_weights(cfg), dtype="f16x3c", device=gpu_device, max_batch=4)
    cfg = NestiConfig()
    net = NestiNet(cfg, weights.synthetic_weights(cfg), dtype="f16x3", device=gpu_device, max_batch=4)
    with pytest.raises(_lib.NestiError, match="not a NESTI_F16X3C model"):
        net.set_gate_margin(0.1)
    with pytest.raises(_lib.NestiError):
        net.cascade_stats()


def test_gate_error_export_import_and_mix_switches(case, gpu_device):
    """The multi-GPU plumbing of the gate's error counter on real kernels (nesti_model_gate_error_export / _import: what rides in
    the spare row of dist.py's all-gather), and the experimental single-product switches: off by default (a model only packs the
    extra copies after nesti_experiment_mix_enable(1)), bit-identical when enabled but switched off, and a measurably different
    -- yet close -- result when a tap layer runs one product instead of three."""
    from nesti_net_amd import _lib
    from nesti_net_amd.model import NestiNet
    cfg, W, pts, q, points, n_eff, ref, n_all, p16, e16, pfl, efl = case
    net = NestiNet(cfg, W, dtype="f16x3c", device=gpu_device, max_batch=B)
    net.set_gate_margin(0.05)
    net.cascade_stats(reset=True)
    net(points, n_eff)
    st = net.cascade_stats()
    buf = torch.zeros(4, dtype=torch.float32, device=gpu_device)
    net.export_gate_error(buf[1:2])
    torch.cuda.synchronize()
    assert buf[1].item() == pytest.approx(st["max_margin_err"], rel=0, abs=0) and buf[0].item() == 0.0
    # another rank has measured a larger error: this model's threshold follows; NaN / inf / negative entries are ignored
    others = torch.tensor([0.01, float("nan"), 2.0 * st["max_margin_err"], float("inf"), -1.0], dtype=torch.float32, device=gpu_device)
    net.import_gate_error(others)
    st2 = net.cascade_stats()
    assert st2["max_margin_err"] == pytest.approx(2.0 * st["max_margin_err"], rel=1e-6)
    assert st2["tau_eff"] == pytest.approx(max(st2["tau"], 1.5 * st2["max_margin_err"]), rel=1e-6)
    net.import_gate_error(torch.tensor([1e-6], dtype=torch.float32, device=gpu_device))      # a smaller value changes nothing
    assert net.cascade_stats()["max_margin_err"] == st2["max_margin_err"]
    with pytest.raises(_lib.NestiError, match="nesti_experiment_mix_enable"):
        net.set_expert_mix(0b001000)                                      # this model was created without the extra copies
    del net
    # ---- the experiment switches -----------------------------------------------------------------------------------------
    lib = _lib.load()
    lib.nesti_experiment_mix_enable(1)
    try:
        x3 = NestiNet(cfg, W, dtype="f16x3", device=gpu_device, max_batch=B)
    finally:
        lib.nesti_experiment_mix_enable(0)
    mups = x3.mups(points, n_eff)
    base = x3.experts(mups, ref[1]).clone()
    assert torch.equal(base, ref[0])                                      # enabled but switched off: f16x3 proper
    x3.set_expert_mix(0b001000)                                           # inception2's 5^3 layer single-product
    mixed = x3.experts(mups, ref[1]).clone()
    x3.set_expert_mix(0)
    assert torch.equal(x3.experts(mups, ref[1]), base)
    a, b = mixed.double().cpu().numpy(), base.double().cpu().numpy()
    omc = 1 - (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))
    print("expert mix 001000: 1-cos p50 %.3g max %.3g" % (np.median(omc), omc.max()))
    assert not np.array_equal(a, b) and np.median(omc) < 1e-6 and omc.max() < 5e-2
    pg, eg = x3.gate(mups)
    assert torch.equal(pg, ref[2])
    x3.set_gate_mix(1)
    pm, em = x3.gate(mups)
    x3.set_gate_mix(0)
    assert not torch.equal(pm, pg) and (pm - pg).abs().max().item() < 0.05 and (em != eg).float().mean().item() < 0.02
    with pytest.raises(_lib.NestiError, match="six bits"):
        x3.set_expert_mix(1 << 6)


def test_cascade_on_the_3_gaussian_grid(gpu_device):
    """--num_gaussians 3 with the two-stage gate: the filter pass's one-tap layers (conv1|conv4 of every block AND the k = 1 conv2
    layers of inception3 / 4, all on the 3^3 grid embedded in 4^3) run the X2 loop; tau = inf must still be f16x3 bit for bit and
    tau = 0 must land within the filter's error of it."""
    from conftest import golden_patch_files, load_golden_patches
    from nesti_net_amd import weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    cfg = NestiConfig(n_gaussians=3, gmm_variance=0.111)
    W = weights.synthetic_weights(cfg)
    g = load_golden_patches([p for p in golden_patch_files() if "ellipsoid20k" in p][0])
    Bq = min(48, len(g["points"]))
    p, n = torch.as_tensor(g["points"][:Bq], device=gpu_device), torch.as_tensor(g["n_eff"][:Bq], device=gpu_device)
    x3 = NestiNet(cfg, W, dtype="f16x3", device=gpu_device, max_batch=Bq)
    want = [t.clone() for t in x3(p, n)]
    net = NestiNet(cfg, W, dtype="f16x3c", device=gpu_device, max_batch=Bq)
    net.set_gate_margin(1e30)
    for a, b in zip(net(p, n), want):
        assert torch.equal(a, b)
    net.set_gate_margin(0.0)
    _, _, probs = net(p, n)
    err = float((probs - want[2]).abs().max().item())
    print("3^3 grid, filter alone: prob err vs f16x3 %.4g" % err)
    assert err < 0.02 and net.cascade_stats()["rechecked"] == Bq        # the tau = inf call above rechecked everything, this one nothing more
