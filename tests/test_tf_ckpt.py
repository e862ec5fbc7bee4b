"""TF1 checkpoint / py2 pickle readers (N1) against files written here at the FORMAT level
(LevelDB table + BundleEntryProto encoders below).  No TensorFlow-written file is available, so
this pins the reader to the published formats, not to TF itself."""
import argparse
import json
import os
import pickle
import struct
import sys
import types

import numpy as np
import pytest


def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _entry(dtype, shape, offset, size):
    dims = b"".join(b"\x12" + _vi(len(_vi(d)) + 1) + b"\x08" + _vi(d) for d in shape)      # dim { size }
    msg = b"\x08" + _vi(dtype) + b"\x12" + _vi(len(dims)) + dims
    msg += b"\x18" + _vi(0) + b"\x20" + _vi(offset) + b"\x28" + _vi(size) + b"\x35" + struct.pack("<I", 0)
    return msg


def _block(items, restart_interval=16):
    buf, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(buf))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        buf += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        prev = k
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def write_bundle(prefix, tensors, per_block=7):
    """Minimal tensor-bundle writer: uncompressed table, several data blocks, one shard."""
    data, items = bytearray(), [(b"", b"\x08\x01")]                       # header entry (empty key)
    for name in sorted(tensors):
        a = np.ascontiguousarray(tensors[name], dtype=np.float32)
        items.append((name.encode(), _entry(1, a.shape, len(data), a.nbytes)))
        data += a.tobytes()
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    out, index_items = bytearray(), []
    for i in range(0, len(items), per_block):
        blk = _block(items[i:i + per_block], restart_interval=3)
        index_items.append((items[min(i + per_block, len(items)) - 1][0], _vi(len(out)) + _vi(len(blk))))
        out += blk + b"\x00" + struct.pack("<I", 0)                         # type + crc trailer
    meta = _block([])
    meta_h = _vi(len(out)) + _vi(len(meta))
    out += meta + b"\x00" + struct.pack("<I", 0)
    idx = _block(index_items, restart_interval=1)
    idx_h = _vi(len(out)) + _vi(len(idx))
    out += idx + b"\x00" + struct.pack("<I", 0)
    footer = (meta_h + idx_h).ljust(40, b"\x00") + struct.pack("<Q", 0xdb4775248b80fb57)
    open(prefix + ".index", "wb").write(bytes(out) + footer)


def _tf_names(W):
    """Rename this package's variables the way TF1 names them (EMA shadows under the moments op)."""
    out = {}
    for k, v in W.items():
        if k.endswith("/bn/mean"):
            sc = k[:-len("/mean")]
            out["%s/%s/moments/Squeeze/ExponentialMovingAverage" % (sc, sc)] = v
        elif k.endswith("/bn/var"):
            sc = k[:-len("/var")]
            out["%s/%s/moments/Squeeze_1/ExponentialMovingAverage" % (sc, sc)] = v
        else:
            out[k] = v
    out["beta1_power"] = np.float32(0.5)                                    # optimizer slots are ignored
    return out


def test_bundle_roundtrip_and_name_mapping(tmp_path):
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import tf_ckpt
    rng = np.random.RandomState(0)
    expected = {"a_conv1/weights": (1, 1, 1, 4, 6), "a_conv1/biases": (6,), "a_conv1/bn/beta": (6,),
                "a_conv1/bn/gamma": (6,), "a_conv1/bn/mean": (6,), "a_conv1/bn/var": (6,),
                "fc4x/weights": (5, 3), "fc4x/biases": (3,)}
    W = {k: rng.randn(*s).astype(np.float32) for k, s in expected.items()}
    for i in range(40):                                                      # force several table blocks
        W["pad%03d/weights" % i] = rng.randn(2, 2).astype(np.float32)
        expected["pad%03d/weights" % i] = (2, 2)
    prefix = str(tmp_path / "model.ckpt")
    write_bundle(prefix, _tf_names(W))
    raw = tf_ckpt.read_bundle(prefix)
    assert "beta1_power" in raw and len(raw) == len(W) + 1
    got = tf_ckpt.map_variables(raw, expected)
    assert set(got) == set(expected)
    for k in expected:
        assert np.array_equal(got[k], W[k]), k
    with pytest.raises(KeyError):
        tf_ckpt.map_variables(raw, {"missing/weights": (1,)})
    with pytest.raises(ValueError):
        tf_ckpt.map_variables(raw, {"fc4x/weights": (3, 5)})
    with pytest.raises(ValueError):
        open(prefix + ".index", "ab").write(b"x")
        tf_ckpt.read_index(prefix + ".index")


def test_py2_pickles_and_full_model_dir(tmp_path, monkeypatch):
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import tf_ckpt, weights
    from nesti_net_amd.config import NestiConfig
    cfg = NestiConfig()
    ns = argparse.Namespace(patch_radius=[0.01, 0.03, 0.05], num_point=512, n_experts=7, num_gaussians=8, gmm_variance=0.0156,
                            model="experts_n_est",
                            expert_loss_type="simple", loss_type="cos",
                            expert_dict=json.dumps({str(k): json.dumps(v) for k, v in cfg.expert_dict.items()}))
    pickle.dump(ns, open(str(tmp_path / "parameters.p"), "wb"), protocol=2)
    mod = types.ModuleType("sklearn.mixture.gaussian_mixture")
    cls = type("GaussianMixture", (object,), {"__module__": "sklearn.mixture.gaussian_mixture"})
    mod.GaussianMixture = cls
    sys.modules["sklearn.mixture.gaussian_mixture"] = mod
    try:
        g = cls()
        g.weights_ = np.ones(512) / 512
        g.means_ = np.zeros((512, 3))
        g.covariances_ = 0.0156 * np.ones((512, 3))
        pickle.dump(g, open(str(tmp_path / "gmm.p"), "wb"), protocol=2)
    finally:
        del sys.modules["sklearn.mixture.gaussian_mixture"]
    cfg2 = tf_ckpt.load_parameters(str(tmp_path / "parameters.p"))
    assert cfg2 == cfg
    # the ablation drivers pickle the same kind of namespace, without expert fields
    # (train_n_est.py:99, train_n_est_w_switching.py:111); --model selects the graph
    for model, radius in (("ss_norm_est", [0.05]), ("ms_norm_est", [0.01, 0.03, 0.05]), ("ms_sw_n_est", [0.01, 0.05])):
        ns_a = argparse.Namespace(patch_radius=radius, num_point=512, num_gaussians=8, gmm_variance=0.0156, model=model)
        pickle.dump(ns_a, open(str(tmp_path / "parameters_a.p"), "wb"), protocol=2)
        got = tf_ckpt.load_parameters(str(tmp_path / "parameters_a.p"))
        want = NestiConfig.for_model(model)
        assert got.arch == want.arch and got.patch_radius == radius and got.n_towers == want.n_towers
        weights.describe(got)          # the graph builder accepts it
    w, mu, cov = tf_ckpt.load_gmm(str(tmp_path / "gmm.p"))
    assert w.shape == (512,) and cov[0, 0] == 0.0156
    # a whole trained-model directory; the real graph's variable list restricted to its small tensors
    # (the conv weights alone are 700 MB) so the CPU suite stays light
    full = weights.describe(cfg)
    assert len(full) == 976
    exp = {k: v for k, v in full.items() if int(np.prod(v)) <= 1 << 17}
    assert len(exp) > 700 and "fc4noise/weights" in exp and "inception1gating_conv_conv1/bn/mean" in exp
    monkeypatch.setattr(weights, "describe", lambda c: exp)
    rng = np.random.RandomState(1)
    W = {k: rng.rand(*s).astype(np.float32) for k, s in exp.items()}
    write_bundle(str(tmp_path / "model.ckpt"), _tf_names(W), per_block=50)
    cfg3, W3 = tf_ckpt.load_reference_model(str(tmp_path) + os.sep)
    assert cfg3 == cfg and list(W3) == list(exp)
    assert all(np.array_equal(W3[k], W[k]) for k in exp)
