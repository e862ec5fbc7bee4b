"""N1 end to end: the reference's trained-model artefacts (parameters.p + gmm.p + model.ckpt.index / .data, written
here at the format level with the EMA shadow variables in both of TF's spellings) -> tf_ckpt.load_reference_model ->
NestiNet -> normals, bit-identical to the model built from the same variables directly
(test_n_est_w_experts.py:46-54, 98-105, 201), and through the command line's model.ckpt branch."""
import os

import numpy as np
import pytest
import torch

from ckpt_writer import write_model_dir

pytestmark = pytest.mark.gpu


def test_reference_artefacts_to_normals(tmp_path, gpu_device):
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import synth, tf_ckpt, weights
    from nesti_net_amd.cli import main
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.pipeline import NormalEstimator
    cfg = NestiConfig()
    W = weights.synthetic_weights(cfg)
    model_dir = str(tmp_path / "my_experts") + os.sep
    os.makedirs(model_dir)
    write_model_dir(model_dir, cfg, W, uniquified=True)
    cfg2, W2 = tf_ckpt.load_reference_model(model_dir)
    assert cfg2 == cfg and list(W2) == list(W)
    assert all(np.array_equal(W2[k], W[k]) for k in W)
    pts = synth.make_cloud("ellipsoid", n=3000, seed=50)[0]
    q = np.arange(0, 3000, 7)
    a = NormalEstimator(cfg, W, dtype="f16", device=gpu_device, batch=300).estimate(pts, pidx=q)
    b = NormalEstimator(cfg2, W2, dtype="f16", device=gpu_device, batch=300).estimate(pts, pidx=q)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    # the command line picks the model.ckpt branch when the directory holds no model.nstw
    data = tmp_path / "pcp"
    data.mkdir()
    np.savetxt(str(data / "shapeA.xyz"), pts, fmt="%.9g")
    np.savetxt(str(data / "shapeA.pidx"), q, fmt="%d")
    (data / "testset.txt").write_text("shapeA\n")
    rc = main(["--results_path", model_dir, "--dataset_name", "synth", "--dataset_path", str(data) + os.sep,
               "--testset", "testset.txt", "--sparse_patches", "1", "--dtype", "f16"])
    assert rc == 0
    out = os.path.join(model_dir, "synth_results")
    assert "model.ckpt" in open(os.path.join(out, "log.txt")).read()
    assert np.array_equal(np.loadtxt(os.path.join(out, "shapeA.normals")), a[0].astype(np.float64))
    assert np.array_equal(np.loadtxt(os.path.join(out, "shapeA.experts")).astype(np.int32), a[1])
