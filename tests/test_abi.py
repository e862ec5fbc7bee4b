"""CPU-only checks of the boundary: the C-ABI library loads without a GPU and exports every
symbol include/nesti_hip.h declares; host-only entry points behave."""
import ctypes
import os
import re

import numpy as np

from conftest import REPO


def _header_symbols():
    src = open(os.path.join(REPO, "include", "nesti_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nesti_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), "missing export %s" % s
        assert s in _lib.SIGNATURES, "no ctypes signature for %s" % s
    assert sorted(_lib.SIGNATURES) == syms


def test_gmm_grid_matches_oracle():
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd.model import get_3d_grid_gmm
    from oracle import mups_ref
    w, mu, sg = get_3d_grid_gmm((8, 8, 8), 0.0156)
    ow, omu, osg = mups_ref.grid_gmm(8, 0.0156)
    assert np.array_equal(w, ow.astype(np.float32))
    assert np.array_equal(mu, omu.astype(np.float32))
    assert np.array_equal(sg, osg.astype(np.float32))
    w3, mu3, _ = get_3d_grid_gmm((3, 3, 3), 0.04)
    o3 = mups_ref.grid_gmm(3, 0.04)
    assert np.array_equal(mu3, o3[1].astype(np.float32)) and np.allclose(w3, 1 / 27.0)


def test_describe_matches_reference_graph():
    """Variable names/shapes follow models/experts_n_est.py (scopes, Python-2 filter counts)."""
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import weights
    from nesti_net_amd.config import NestiConfig
    d = weights.describe(NestiConfig())
    assert d["inception1gating_conv_conv1/weights"] == (1, 1, 1, 60, 128)
    assert d["inception3gating_conv_conv3/weights"] == (5, 5, 5, 256, 128)
    assert d["inception8gating_conv_conv2/weights"] == (1, 1, 1, 512, 256)
    assert d["inception8gating_conv_conv3/weights"] == (2, 2, 2, 512, 256)
    assert d["fc1noise/weights"] == (1536, 1024) and d["fc4noise/weights"] == (128, 7)
    assert "fc4noise/bn/beta" not in d and "fc3noise/bn/beta" in d
    assert d["inception1Expert_0_conv1/weights"] == (1, 1, 1, 20, 128)
    assert d["inception1Expert_6_conv1/weights"] == (1, 1, 1, 60, 42)        # 128/3 under py2 (experts_n_est.py:254)
    assert d["inception1Expert_6_conv2/weights"] == (3, 3, 3, 42, 21)
    assert d["inception2Expert_6_conv1/weights"] == (1, 1, 1, 126, 256)
    assert d["inception6Expert_3_conv3/weights"] == (4, 4, 4, 512, 256)
    assert d["fc4Expert_5/weights"] == (64, 3)
    n_params = sum(int(np.prod(v)) for k, v in d.items() if k.endswith("/weights"))
    assert abs(n_params - 178.3e6) < 0.1e6


def test_describe_switching_model_graph():
    """ms_sw_n_est (models/ms_sw_n_est.py:41-215): three 'ss' towers with scope suffixes noise / small / large."""
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import _lib, weights
    from nesti_net_amd.config import ARCH_SWITCH, NestiConfig
    cfg = NestiConfig.for_model("ms_sw_n_est")
    assert cfg.arch == ARCH_SWITCH and cfg.n_towers == 2 and cfg.n_gate_out == 1
    d = weights.describe(cfg)
    for sfx in ("noise", "small", "large"):
        assert d["inception1%s_conv1/weights" % sfx] == (1, 1, 1, 20, 128)
        assert d["inception3%s_conv3/weights" % sfx] == (5, 5, 5, 256, 128)
        assert d["inception6%s_conv3/weights" % sfx] == (5, 5, 5, 512, 256)
        assert d["fc1%s/weights" % sfx] == (12288, 1024)
        assert "fc4%s/bn/beta" % sfx not in d
    assert d["fc4noise/weights"] == (128, 1) and d["fc4small/weights"] == (128, 3) and d["fc4large/weights"] == (128, 3)
    assert not any("inception4" in k or "inception7" in k for k in d)          # 4 and 7 are the max-pools
    lib = _lib.load()
    bad = NestiConfig.for_model("ms_sw_n_est")
    bad.patch_radius = [0.01, 0.03, 0.05]
    n = ctypes.c_int(0)
    c = bad.to_c()
    assert lib.nesti_model_describe(ctypes.byref(c), ctypes.byref(n), None, 0) != 0
    assert b"two scales" in lib.nesti_last_error()


def test_config_expert_dict_and_errors():
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import _lib
    from nesti_net_amd.config import NestiConfig
    cfg = NestiConfig()
    c = cfg.to_c()
    assert list(c.expert_scale_lo)[:7] == [0, 0, 1, 1, 2, 2, 0] and list(c.expert_scale_cnt)[:7] == [1, 1, 1, 1, 1, 1, 3]
    assert NestiConfig(expert_dict=None).default_expert_dict() == {0: [0], 1: [0], 2: [1], 3: [1], 4: [2], 5: [2],
                                                                    6: [0, 1, 2]}
    lib = _lib.load()
    bad = NestiConfig(n_gaussians=5).to_c()
    n = ctypes.c_int(0)
    assert lib.nesti_model_describe(ctypes.byref(bad), ctypes.byref(n), None, 0) != 0
    assert b"8^3" in lib.nesti_last_error()
    bad = NestiConfig.for_model("ss_norm_est")
    bad.n_gaussians = 3                      # the ablation models exist for the 8^3 grid only
    c = bad.to_c()
    assert lib.nesti_model_describe(ctypes.byref(c), ctypes.byref(n), None, 0) != 0


def test_describe_3_gaussian_grid_graph():
    """--num_gaussians 3 (27 Gaussians): conv_net_3g for the gate and every expert (models/experts_n_est.py:162-163,
    217-240, 275-276); the expert towers carry the '_expert_conv' scope suffix and ignore the filter divider."""
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import weights
    from nesti_net_amd.config import NestiConfig
    d = weights.describe(NestiConfig(n_gaussians=3, gmm_variance=0.111))
    assert d["inception1gating_conv_conv1/weights"] == (1, 1, 1, 60, 128)
    assert d["inception1gating_conv_conv2/weights"] == (2, 2, 2, 128, 64)
    assert d["inception2gating_conv_conv3/weights"] == (3, 3, 3, 256, 128)
    assert d["inception3gating_conv_conv2/weights"] == (1, 1, 1, 256, 128)
    assert d["inception4gating_conv_conv3/weights"] == (2, 2, 2, 512, 256)
    assert d["fc1noise/weights"] == (12288, 1024) and d["fc4noise/weights"] == (128, 7)
    assert d["inception1Expert_6_expert_conv_conv1/weights"] == (1, 1, 1, 60, 128)     # no divider on this branch
    assert d["inception1Expert_2_expert_conv_conv1/weights"] == (1, 1, 1, 20, 128)
    assert d["fc1Expert_2/weights"] == (12288, 512) and d["fc4Expert_2/weights"] == (64, 3)
    assert not any(k.startswith("inception5") for k in d)


def test_weight_container_roundtrip(tmp_path):
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import weights
    from nesti_net_amd.config import NestiConfig
    cfg = NestiConfig()
    W = {"a/weights": np.arange(24, dtype=np.float32).reshape(1, 1, 1, 4, 6), "a/biases": np.ones(6, np.float32)}
    p = str(tmp_path / "w.nstw")
    weights.save(p, W, cfg)
    W2, cfg2 = weights.load(p)
    assert list(W2) == list(W) and all(np.array_equal(W[k], W2[k]) for k in W)
    assert cfg2 == cfg


def test_python_enums_follow_the_header():
    """config.DTYPES / ARCH_* are the header's enumerators (include/nesti_hip.h), by name and value."""
    import re
    from nesti_net_amd import config
    text = open(os.path.join(REPO, "include", "nesti_hip.h")).read()
    enums = {}
    for body in re.findall(r"enum\s*\{([^}]*)\}", text):
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        for name, val in re.findall(r"(NESTI_[A-Z0-9_]+)\s*=\s*(\d+)", body):
            enums[name] = int(val)
    want = {"f32": "NESTI_F32", "bf16": "NESTI_BF16", "f16": "NESTI_F16", "bf16x3": "NESTI_BF16X3", "f16x3": "NESTI_F16X3",
            "f16x3c": "NESTI_F16X3C", "f16x8": "NESTI_F16X8", "f16x8c": "NESTI_F16X8C"}
    assert set(config.DTYPES) == set(want)
    for k, name in want.items():
        assert config.DTYPES[k] == enums[name], k
    for py, name in (("ARCH_EXPERTS", "NESTI_ARCH_EXPERTS"), ("ARCH_SINGLE", "NESTI_ARCH_SINGLE"),
                     ("ARCH_MULTI", "NESTI_ARCH_MULTI"), ("ARCH_SWITCH", "NESTI_ARCH_SWITCH")):
        assert getattr(config, py) == enums[name], py


def test_tower_workspace_bytes_from_the_configuration_alone():
    """nesti_tower_workspace_bytes needs no device: the pair modes keep two planes per 64-channel group (2x the 16-bit
    footprint), the f16x3c gate figure is the f16 filter pass's, and sizes scale linearly with the batch."""
    import ctypes
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import _lib
    from nesti_net_amd.config import DTYPES, NestiConfig
    lib, c = _lib.load(), NestiConfig().to_c()

    def ws(dtype, tower, batch=1024):
        return lib.nesti_tower_workspace_bytes(ctypes.byref(c), DTYPES[dtype], tower, batch)

    for tower in (-1, 0, 6):
        assert ws("f16", tower) > 0 and ws("f16x3", tower) == 2 * ws("f16", tower) and ws("bf16", tower) == ws("f16", tower)
        assert ws("f16", tower, 2048) == 2 * ws("f16", tower, 1024)
    assert ws("f16x3c", -1) == ws("f16", -1) and ws("f16x3c", 0) == ws("f16x3", 0)
    # the FP8 cross-term modes add the side buffers of e4m3 planes to the expert towers (2 bytes per conv1 channel and voxel of the two
    # 8^3 blocks; they share memory with later buffers where lifetimes allow), nothing to the gating net
    assert ws("f16x8c", -1) == ws("f16", -1) and ws("f16x8", -1) == ws("f16x3", -1)
    for tower in (0, 6):
        assert ws("f16x3", tower) <= ws("f16x8", tower) <= ws("f16x3", tower) + 1024 * 512 * (128 + 256) * 2 and ws("f16x8c", tower) == ws("f16x8", tower)
    assert 1.5e6 < ws("f16", -1) / 1024 < 2.5e6                      # ~2 MB per query for the gating net in 16-bit
    assert ws("f16", 7) == 0 and ws("f16", -2) == 0 and ws("f16", 0, 0) == 0
    # the whole arena of a fused-call batch from the configuration alone (cli.fit_batch): positive, monotone in the batch, the pair
    # modes above the plain ones, the FP8 cross-term modes no smaller than their f16x3 counterparts
    def arena(dtype, batch=4096):
        return lib.nesti_estimate_workspace_bytes_for_config(ctypes.byref(c), DTYPES[dtype], batch)
    assert 0 < arena("f16") < arena("f16x3") and arena("f16x3c", 32768) < arena("f16x3", 32768)   # recheck rounds on a quarter of a large batch
    assert arena("f16x3") <= arena("f16x8") and arena("f16x3c") <= arena("f16x8c")
    assert arena("f16", 8192) > arena("f16", 4096) and arena("f16", 0) == 0


def test_x8_dtypes_are_refused_where_they_do_not_apply():
    """NESTI_F16X8 / NESTI_F16X8C: experts_n_est on the 8^3 grid only -- refused at the argument check (before any device call) for
    the single-tower models and for the 3^3 grid; the setters refuse a null model."""
    import ctypes
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import _lib
    from nesti_net_amd.config import ARCH_SINGLE, DTYPES, NestiConfig
    lib = _lib.load()
    h = ctypes.c_void_p()
    dummy = (_lib.CTensor * 1)()
    for cfg in (NestiConfig(patch_radius=[0.05], n_experts=1, expert_dict={0: [0]}, arch=ARCH_SINGLE), NestiConfig(n_gaussians=3)):
        c = cfg.to_c()
        for dt in ("f16x8", "f16x8c"):
            assert lib.nesti_model_create(ctypes.byref(c), dummy, 0, DTYPES[dt], ctypes.byref(h)) != 0
            msg = lib.nesti_last_error().decode()
            assert "F16X8" in msg or "two-stage gate" in msg, msg
    assert lib.nesti_model_set_x8_layers(None, 0xF) != 0 and lib.nesti_model_set_x8_guard(None, ctypes.c_float(0.1)) != 0


def test_e2m3_encoder_rounds_to_nearest_even_and_saturates():
    """The host-side FP6 encoder of the weight packing (model.hip: host_f32_to_e2m3) against the format's definition: every code decodes and
    encodes back to itself, a value between two grid points goes to the nearer one, an exact midpoint to the even mantissa, everything
    beyond 7.5 saturates, the sign rides in bit 5, the scale divides."""
    from nesti_net_amd import _lib
    lib = _lib.load()

    def dec(c):
        e, m = (c >> 3) & 3, c & 7
        v = m * 0.125 if e == 0 else (1 + m * 0.125) * 2.0 ** (e - 1)
        return -v if c & 32 else v

    grid = [dec(c) for c in range(32)]
    assert grid == sorted(grid) and grid[31] == 7.5 and grid[1] == 0.125
    for c in range(64):
        if c == 32:
            continue                                        # -0 encodes as the sign bit alone: checked below
        assert lib.nesti_f32_to_e2m3(dec(c), 1.0) == c
    assert lib.nesti_f32_to_e2m3(-0.0, 1.0) in (0, 32)
    for c in range(31):
        lo, hi = grid[c], grid[c + 1]
        assert lib.nesti_f32_to_e2m3(lo + 0.25 * (hi - lo), 1.0) == c and lib.nesti_f32_to_e2m3(lo + 0.75 * (hi - lo), 1.0) == c + 1
        assert lib.nesti_f32_to_e2m3(0.5 * (lo + hi), 1.0) == (c if c % 2 == 0 else c + 1)       # ties to the even mantissa
        assert lib.nesti_f32_to_e2m3(-0.5 * (lo + hi), 1.0) == 32 + (c if c % 2 == 0 else c + 1)
    for v in (7.6, 7.75, 8.0, 100.0, 1e30, float("inf")):
        assert lib.nesti_f32_to_e2m3(v, 1.0) == 31 and lib.nesti_f32_to_e2m3(-v, 1.0) == 63
    assert lib.nesti_f32_to_e2m3(0.06, 1.0) == 0 and lib.nesti_f32_to_e2m3(0.0626, 1.0) == 1     # the probe's cases (profiles/r06_fp6_probe.txt)
    assert lib.nesti_f32_to_e2m3(24.0, 0.25) == lib.nesti_f32_to_e2m3(6.0, 1.0) == 28
