"""MuPS HIP kernel (through the C-ABI) vs the fp64 oracle (oracle/mups_ref.py)."""
import numpy as np
import pytest
import torch

from conftest import golden_patch_files, load_golden_patches

pytestmark = pytest.mark.gpu

TOL_F32 = 5e-6     # abs, on L2-normalised channels (|v| <= 1); fp32 kernel vs fp64 restatement
TOL_BF16 = 6e-3    # bf16 storage: 2^-8 relative on |v| <= 1


def _run(cfg, pts, n_eff, dev, dtype="f32", cstride=None):
    from nesti_net_amd.model import mups_forward
    out = mups_forward(cfg, torch.as_tensor(pts, device=dev), torch.as_tensor(n_eff, device=dev), out_dtype=dtype,
                       out_cstride=cstride)
    torch.cuda.synchronize()
    return out.float().cpu().numpy()


def test_mups_matches_oracle_on_reference_patches(gpu_device):
    from nesti_net_amd.config import NestiConfig
    from oracle import mups_ref
    path = [p for p in golden_patch_files() if "ellipsoid100k" in p][0]
    g = load_golden_patches(path)
    cfg = NestiConfig()
    pts, n_eff = g["points"][:12], g["n_eff"][:12]
    got = _run(cfg, pts, n_eff, gpu_device)
    ref = mups_ref.mups_assemble(pts, n_eff, 3)
    err = np.abs(got - ref).max()
    print("mups f32 max abs err", err)
    assert got.shape == (12, 8, 8, 8, 60)
    assert err < TOL_F32


def test_mups_edge_cases(gpu_device):
    """n_eff = P (no masked rows), n_eff = 1, tiny n_eff, non-zero garbage in masked rows,
    and the zero-padded batch tail (n_eff = 0 -> zeros, not NaN)."""
    from nesti_net_amd.config import NestiConfig
    from oracle import mups_ref
    cfg = NestiConfig()
    rng = np.random.RandomState(5)
    S, P = 3, 512
    n_eff = np.array([[512, 512, 512], [1, 2, 3], [511, 510, 64], [65, 63, 129], [0, 0, 0], [7, 0, 300]], np.int32)
    B = len(n_eff)
    pts = (rng.uniform(-1, 1, (B, S * P, 3)) * 0.7).astype(np.float32)
    for b in range(B):
        for s in range(S):
            pts[b, s * P + n_eff[b, s]:(s + 1) * P] = 0      # reference zero padding
    pts[3, 0 * P + 67:0 * P + 80] = 0.25                       # rows > n_eff are masked whatever they hold
    got = _run(cfg, pts, n_eff, gpu_device)
    assert np.all(np.isfinite(got))
    ref = mups_ref.mups_assemble(pts, np.maximum(n_eff, 1), 3)
    for b in range(B):
        for s in range(S):
            sl = slice(20 * s, 20 * s + 20)
            if n_eff[b, s] == 0:
                assert np.all(got[b, ..., sl] == 0)
            else:
                assert np.abs(got[b, ..., sl] - ref[b, ..., sl]).max() < TOL_F32, (b, s)


def test_mups_padded_bf16_layout(gpu_device):
    from nesti_net_amd.config import NestiConfig
    from oracle import mups_ref
    g = load_golden_patches([p for p in golden_patch_files() if "ellipsoid20k" in p][0])
    cfg = NestiConfig()
    pts, n_eff = g["points"][:6], g["n_eff"][:6]
    ref = mups_ref.mups_assemble(pts, n_eff, 3)
    for dt in ("bf16", "f16"):
        got = _run(cfg, pts, n_eff, gpu_device, dtype=dt, cstride=64)
        assert got.shape == (6, 8, 8, 8, 64)
        assert np.all(got[..., 60:] == 0)
        assert np.abs(got[..., :60] - ref).max() < TOL_BF16


def test_mups_single_scale_config(gpu_device):
    """BASELINE config 0 shape: one radius."""
    from nesti_net_amd.config import NestiConfig
    from oracle import mups_ref
    g = load_golden_patches([p for p in golden_patch_files() if "sphere8k" in p][0])
    cfg = NestiConfig(patch_radius=[0.05], n_experts=1, expert_dict={0: [0]})
    got = _run(cfg, g["points"][:8], g["n_eff"][:8], gpu_device)
    ref = mups_ref.mups_assemble(g["points"][:8], g["n_eff"][:8], 1)
    assert got.shape == (8, 8, 8, 8, 20)
    assert np.abs(got - ref).max() < TOL_F32


def test_mups_properties_full_size(gpu_device):
    """Size-independent properties at the BASELINE batch scale (4096 queries x 3 x 512):
    unit L2 norm per channel, permutation invariance over the valid rows."""
    from nesti_net_amd.config import NestiConfig
    cfg = NestiConfig()
    rng = np.random.RandomState(9)
    B, S, P = 4096, 3, 512
    n_eff = rng.randint(1, P + 1, size=(B, S)).astype(np.int32)
    pts = (rng.normal(size=(B, S * P, 3)) * 0.35).astype(np.float32)
    mask = (np.arange(P)[None, None, :] < n_eff[:, :, None]).reshape(B, S * P)
    pts *= mask[..., None]
    dev = gpu_device
    from nesti_net_amd.model import mups_forward
    a = mups_forward(cfg, torch.as_tensor(pts, device=dev), torch.as_tensor(n_eff, device=dev))
    norms = (a.double() ** 2).sum(dim=(1, 2, 3))          # [B, 60]
    assert torch.all((norms - 1).abs() < 1e-4)
    # permute the valid rows of scale 1 of every query: the statistics are symmetric functions
    pts2 = pts.copy()
    for b in range(0, B, 64):
        m = n_eff[b, 1]
        perm = rng.permutation(m)
        pts2[b, P:P + m] = pts[b, P:P + m][perm]
    b2 = mups_forward(cfg, torch.as_tensor(pts2, device=dev), torch.as_tensor(n_eff, device=dev))
    assert (a - b2).abs().max().item() < 1e-5


def test_mups_other_patch_sizes_and_scale_counts(gpu_device):
    """P = 64 points per scale, 2 and 4 scales (the kernel is generic in S <= 4 and P)."""
    from nesti_net_amd.config import NestiConfig
    from oracle import mups_ref
    rng = np.random.RandomState(11)
    for S, P in ((2, 64), (4, 96)):
        cfg = NestiConfig(patch_radius=[0.01 * (i + 1) for i in range(S)], num_point=P, n_experts=1,
                          expert_dict={0: list(range(S))})
        B = 5
        n_eff = rng.randint(1, P + 1, size=(B, S)).astype(np.int32)
        n_eff[0, :] = P
        pts = (rng.normal(size=(B, S * P, 3)) * 0.3).astype(np.float32)
        mask = (np.arange(P)[None, None, :] < n_eff[:, :, None]).reshape(B, S * P)
        pts *= mask[..., None]
        got = _run(cfg, pts, n_eff, gpu_device)
        ref = mups_ref.mups_assemble(pts, n_eff, S)
        assert got.shape == (B, 8, 8, 8, 20 * S)
        assert np.abs(got - ref).max() < TOL_F32, (S, P)
