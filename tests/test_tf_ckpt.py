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

from ckpt_writer import tf_names, write_bundle


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
    write_bundle(prefix, tf_names(W))
    raw = tf_ckpt.read_bundle(prefix)
    assert "beta1_power" in raw and len(raw) == len(W) + 1
    got = tf_ckpt.map_variables(raw, expected)
    assert set(got) == set(expected)
    for k in expected:
        assert np.array_equal(got[k], W[k]), k
    # the spelling with re-entered name scopes: <scope>/bn/<scope>/bn_1/moments/Squeeze_1/ExponentialMovingAverage_1
    prefix_u = str(tmp_path / "model_u.ckpt")
    named = tf_names(W, uniquified=True)
    assert any(k.endswith("ExponentialMovingAverage_1") and "/bn_1/" in k for k in named)
    write_bundle(prefix_u, named)
    got_u = tf_ckpt.map_variables(tf_ckpt.read_bundle(prefix_u), expected)
    assert all(np.array_equal(got_u[k], W[k]) for k in expected)
    # two shadow variables of the same kind under one scope cannot be told apart: an error, not a guess
    dup = dict(raw)
    dup["a_conv1/bn/a_conv1/bn_1/moments/Squeeze/ExponentialMovingAverage"] = raw["a_conv1/bn/a_conv1/bn/moments/Squeeze/ExponentialMovingAverage"]
    with pytest.raises(KeyError):
        tf_ckpt.map_variables(dup, expected)
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
    write_bundle(str(tmp_path / "model.ckpt"), tf_names(W), per_block=50)
    cfg3, W3 = tf_ckpt.load_reference_model(str(tmp_path) + os.sep)
    assert cfg3 == cfg and list(W3) == list(exp)
    assert all(np.array_equal(W3[k], W[k]) for k in exp)
