"""N1 end to end: the reference's trained-model artefacts (parameters.p + gmm.p + model.ckpt.index / .data, written
here at the format level with the EMA shadow variables in both of TF's spellings) -> tf_ckpt.load_reference_model ->
NestiNet -> normals, bit-identical to the model built from the same variables directly
(test_n_est_w_experts.py:46-54, 98-105, 201), and through the command line's model.ckpt branch.

The restored model is also held to the ORACLE, not only to itself: the f32 mode built from the restored
variables against oracle.net_ref.moe_forward fed with the SAME restored variables and configuration (arg-max exact outside
parity.TIE_MARGIN, 1 - cos <= 1e-5).  The reader stays pinned at the FORMAT level (tests/test_tf_ckpt.py) until a file written
by TensorFlow itself exists: none ships with the reference and TF 1.12 cannot be installed here."""
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
    pts = synth.make_cloud("ellipsoid", n=3000, seed=50)[0]
    q = np.arange(0, 3000, 7)
    # a gate that spreads its arg-max over the experts, so that the oracle comparison below exercises every restored tower
    from nesti_net_amd.calibrate import calibrate_gate
    from nesti_net_amd.provider import CloudPatches
    cp = CloudPatches(pts, cfg, device=gpu_device, pidx=q)
    p_d, n_d = cp.build(0, len(q))
    W = calibrate_gate(cfg, W, p_d, n_d, device=gpu_device)
    model_dir = str(tmp_path / "my_experts") + os.sep
    os.makedirs(model_dir)
    write_model_dir(model_dir, cfg, W, uniquified=True)
    cfg2, W2 = tf_ckpt.load_reference_model(model_dir)
    assert cfg2 == cfg and list(W2) == list(W)
    assert all(np.array_equal(W2[k], W[k]) for k in W)
    # ---- restored variables -> f32 mode vs the fp64 oracle on the same restored variables ---------------------------------
    from nesti_net_amd import parity
    from nesti_net_amd.model import NestiNet
    from oracle import mups_ref, net_ref, patches_ref
    rows = q[:64]
    o_pts, o_neff, _, _ = patches_ref.extract_patches(pts, rows, cp.r_abs, cfg2.num_point, cp.seed)
    assert np.array_equal(p_d[:64].cpu().numpy().view(np.uint32), o_pts.view(np.uint32))
    mups_o = mups_ref.mups_assemble(o_pts, o_neff, cfg2.n_scales)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ref = net_ref.over_chunks(lambda sl: net_ref.moe_forward(mups_o[sl], W2, expert_dict=cfg2.expert_dict, dtype=torch.float64, top1_only=True), 64)
    ref = {k: torch.cat([r[k] for r in ref]).numpy() for k in ("probs", "expert", "normals")}
    n32, e32, p32 = NestiNet(cfg2, W2, dtype="f32", device=gpu_device, max_batch=64)(p_d[:64], n_d[:64])
    torch.cuda.synchronize()
    srt = np.sort(ref["probs"], axis=1)
    agree = e32.cpu().numpy() == ref["expert"]
    assert np.all(agree | (srt[:, -1] - srt[:, -2] < parity.TIE_MARGIN))
    assert np.abs(p32.cpu().numpy() - ref["probs"]).max() <= parity.F32_PROB_ERR_BOUND
    a32, b32 = n32.cpu().numpy().astype(np.float64)[agree], ref["normals"][agree]
    omc = 1 - (a32 * b32).sum(1) / (np.linalg.norm(a32, axis=1) * np.linalg.norm(b32, axis=1))
    print("restored model vs oracle: experts used", np.unique(ref["expert"]).tolist(), "1-cos max", omc.max())
    assert len(np.unique(ref["expert"])) >= 4 and np.all(omc <= 1e-5)
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
