"""The pair modes 'f16x3' / 'bf16x3' (NESTI_F16X3 / NESTI_BF16X3): activations and weights as 16-bit (hi, lo) pairs,
three MFMA products per multiply through the pair K loop of the f16 / bf16 kernels (planes [hi | lo] per 64-channel group,
weight rows [W_hi | W_lo] per K chunk; f16x3 also scales each layer's weights by a power of two into f16's normal range).

f16x3 is the mode that holds test_n_est_w_experts.py's outputs to the north star's tolerance (arg-max exact or
margin-flagged, normals within 1e-5 cosine) without the fp32 MFMA rate; bf16x3 (2^-17 operands) sits at the edge of
it.  Checked here against the fp64 CPU oracle on golden patches, and against the exact-fp32 mode on every graph the
builder makes (each exercises a different part of the plane layout: flattened FC inputs, the 3^3 grid's embedded rows
and max-pool, two 64-channel groups of MuPS channels, first-block widths that are not multiples of 64).  The
10 240-query and 100 000-query figures are in tests/test_gpu_fixtures.py and bench.py; other clouds in
scripts/pair_mode_sweep.py."""
import numpy as np
import pytest
import torch

from conftest import golden_patch_files, load_golden_patches

pytestmark = pytest.mark.gpu

COS_TOL = 1e-5
MODES = ["f16x3", "bf16x3"]
TORCH_DT = {"f16x3": torch.float16, "bf16x3": torch.bfloat16}


def _cos(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def _golden(name, rows):
    g = load_golden_patches([p for p in golden_patch_files() if name in p][0])
    return g["points"][:rows], g["n_eff"][:rows]


@pytest.mark.parametrize("mode", MODES)
def test_pair_mode_matches_the_fp64_oracle_on_golden_patches(mode, gpu_device):
    from nesti_net_amd import parity, weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    from oracle import mups_ref, net_ref
    cfg = NestiConfig()
    W = weights.synthetic_weights(cfg)
    pa, na = _golden("ellipsoid100k", 10)
    pb, nb = _golden("ellipsoid20k", 6)
    pts, n_eff = np.concatenate([pa, pb]), np.concatenate([na, nb])
    mups_o = mups_ref.mups_assemble(pts, n_eff, 3)
    full = net_ref.moe_forward(mups_o[:6], W, dtype=torch.float64, top1_only=False)
    top1 = net_ref.moe_forward(mups_o, W, dtype=torch.float64, top1_only=True)
    net = NestiNet(cfg, W, dtype=mode, device=gpu_device, max_batch=16)
    p, n = torch.as_tensor(pts, device=gpu_device), torch.as_tensor(n_eff, device=gpu_device)
    # the MuPS rows the towers read: two planes per 64-channel group, hi + lo = the fp32 value to 2^-17 (bf16 pairs)
    mups = net.mups(p, n)
    assert mups.shape == (16, 8, 8, 8, 128) and mups.dtype == TORCH_DT[mode]
    m = mups.float().cpu().numpy()
    assert np.abs(m[..., 0:60] + m[..., 64:124] - mups_o).max() < (2e-5 if mode == "bf16x3" else 5e-6)
    assert not m[..., 60:64].any() and not m[..., 124:128].any()
    normals, expert, probs = net(p, n)
    torch.cuda.synchronize()
    pe = np.abs(probs.cpu().numpy() - top1["probs"].numpy()).max()
    srt = np.sort(top1["probs"].numpy(), axis=1)
    margin = srt[:, -1] - srt[:, -2]
    agree = expert.cpu().numpy() == top1["expert"].numpy()
    c = _cos(normals.cpu().numpy()[agree], top1["normals"].numpy()[agree])
    n_est = net.experts(mups[:6], None).cpu().numpy()
    ca = _cos(n_est, full["n_est"].numpy())
    print(mode, "vs oracle: prob err", pe, "agree", agree.mean(), "1-cos max", (1 - c).max(), "all experts", (1 - ca).max())
    assert pe < (1e-3 if mode == "bf16x3" else 1e-4)  # 2^-17 operands on O(10) logits; f16 pairs: like the fp32 mode
    assert np.all(agree | (margin < (2e-3 if mode == "bf16x3" else parity.TIE_MARGIN)))
    assert np.all(1 - c < COS_TOL) and np.all(1 - ca < COS_TOL)
    if mode == "f16x3":
        assert (1 - c).max() < 1e-7 and (1 - ca).max() < 1e-7


def _cases():
    from nesti_net_amd.config import NestiConfig
    yield "experts_3cubed_grid", NestiConfig(n_gaussians=3, gmm_variance=0.111), "ellipsoid20k", None
    yield "ss_norm_est", NestiConfig.for_model("ss_norm_est"), "sphere8k", None
    yield "ms_norm_est", NestiConfig.for_model("ms_norm_est"), "ellipsoid20k", None
    yield "ms_sw_n_est", NestiConfig.for_model("ms_sw_n_est"), "ellipsoid20k", (0, 2)
    yield "four_experts_mixed_scales", NestiConfig(n_experts=4, expert_dict={0: [0], 1: [1, 2], 2: [2], 3: [0, 1, 2]}), "ellipsoid20k", None


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["experts_3cubed_grid", "ss_norm_est", "ms_norm_est", "ms_sw_n_est", "four_experts_mixed_scales"])
def test_pair_mode_agrees_with_the_fp32_mode_on_every_graph(name, mode, gpu_device):
    from nesti_net_amd import weights
    from nesti_net_amd.model import NestiNet
    _, cfg, golden, scales = [c for c in _cases() if c[0] == name][0]
    W = weights.synthetic_weights(cfg)
    B = 9
    pts, n_eff = _golden(golden, B)
    if scales is not None:      # the two-scale switching model: radii 0.01 and 0.05 of the three-scale fixture
        pts = np.concatenate([pts[:, 512 * s:512 * (s + 1)] for s in scales], axis=1)
        n_eff = np.stack([n_eff[:, s] for s in scales], axis=1)
    elif cfg.n_scales == 1 and n_eff.ndim == 2 and n_eff.shape[1] > 1:
        pts, n_eff = pts[:, :512], n_eff[:, :1]
    p, n = torch.as_tensor(pts, device=gpu_device), torch.as_tensor(n_eff, device=gpu_device)
    outs = {}
    for dt in ("f32", mode):
        net = NestiNet(cfg, W, dtype=dt, device=gpu_device, max_batch=B)
        normals, expert, probs = net(p, n)
        mups = net.mups(p, n)
        every = net.experts(mups, None)            # every tower on every row
        torch.cuda.synchronize()
        outs[dt] = (normals.cpu().numpy(), None if expert is None else expert.cpu().numpy(),
                    None if probs is None else probs.cpu().numpy(), every.cpu().numpy())
        del net
    (n_r, e_r, p_r, a_r), (n_t, e_t, p_t, a_t) = outs["f32"], outs[mode]
    worst = (1 - _cos(a_t, a_r)).max()
    print(name, mode, "all towers 1-cos max", worst, "prob err", None if p_r is None else np.abs(p_t - p_r).max())
    assert worst < COS_TOL
    if p_r is not None:
        assert np.abs(p_t - p_r).max() < 3e-4
    same = np.ones(B, bool) if e_r is None else e_t == e_r
    assert same.mean() >= 0.85                     # a flip needs a near-tie; these few rows have none to speak of
    assert np.all(1 - _cos(n_t[same], n_r[same]) < COS_TOL)


@pytest.mark.parametrize("mode", MODES)
def test_pair_mode_limits_two_channel_groups_and_ragged_batches(mode, gpu_device):
    """NESTI_MAX_SCALES = 4: 80 MuPS channels = two 64-channel groups of planes; a batch that is not a multiple of the
    four points conv8n_kernel (or the sixteen conv4n_kernel) handles per workgroup; rows with n_eff = 0 are skipped, not computed."""
    from nesti_net_amd import weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    cfg = NestiConfig(patch_radius=[0.01, 0.02, 0.04, 0.06], num_point=128, n_experts=8, expert_dict=None)
    cfg.expert_dict = cfg.default_expert_dict()
    W = weights.synthetic_weights(cfg)
    rng = np.random.RandomState(5)
    B = 7
    pts = (rng.rand(B, 4 * 128, 3).astype(np.float32) - 0.5) * 1.6
    n_eff = rng.randint(20, 129, size=(B, 4)).astype(np.int32)
    for b in range(B):
        for s in range(4):
            pts[b, 128 * s + n_eff[b, s]:128 * (s + 1)] = 0
    p, n = torch.as_tensor(pts, device=gpu_device), torch.as_tensor(n_eff, device=gpu_device)
    ref = NestiNet(cfg, W, dtype="f32", device=gpu_device, max_batch=B)
    net = NestiNet(cfg, W, dtype=mode, device=gpu_device, max_batch=B)
    assert net.mups_cstride == 2 * 128
    m = net.mups(p, n).float().cpu().numpy()
    m32 = ref.mups(p, n).cpu().numpy()
    for g in range(2):
        hi, lo = m[..., 128 * g:128 * g + 64], m[..., 128 * g + 64:128 * g + 128]
        assert np.abs(hi + lo - m32[..., 64 * g:64 * g + 64]).max() < 1e-6
    a_t = net.experts(net.mups(p, n), None).cpu().numpy()
    a_r = ref.experts(ref.mups(p, n), None).cpu().numpy()
    assert a_t.shape == (8, B, 3)
    assert np.all(1 - _cos(a_t, a_r) < COS_TOL)
    n_t, e_t, p_t = net(p, n)
    n_r, e_r, p_r = ref(p, n)
    assert np.abs(p_t.cpu().numpy() - p_r.cpu().numpy()).max() < 3e-4
    same = (e_t == e_r).cpu().numpy()
    assert np.all(1 - _cos(n_t.cpu().numpy()[same], n_r.cpu().numpy()[same]) < COS_TOL)
