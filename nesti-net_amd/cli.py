"""Command-line drop-in for the reference's ``test_n_est_w_experts.py``: same flags
(``:19-29``), same inputs (``<dataset_path>/<testset>`` shape list, ``<shape>.xyz``, optional
``<shape>.pidx``) and same outputs (``<results_path>/<dataset_name>_results/<shape>.normals``,
``.experts``, ``.experts_probs`` written with ``np.savetxt`` like ``:182-188``, plus ``log.txt``).

The trained-model directory holds either ``model.nstw`` (variables + hyper-parameters, see
:mod:`.weights`) or the reference's own ``parameters.p`` / ``gmm.p`` / ``model.ckpt.*`` (read by
:mod:`.tf_ckpt` without TensorFlow); ``--synthetic_weights`` substitutes seeded random weights (no
checkpoint ships with the reference)."""
import argparse
import os
import sys

import torch

from . import textio
from . import weights as wts
from .config import ARCH_EXPERTS, ARCH_MULTI, ARCH_SINGLE, ARCH_SWITCH, NestiConfig
from .pipeline import NormalEstimator
from .provider import PointcloudPatchDataset


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--results_path", default="log/my_experts/", help="path to trained model, default log/my_experts/")
    p.add_argument("--model", default="experts_n_est", help="Model name [default: experts_n_est]")
    p.add_argument("--dataset_name", type=str, default="pcpnet", help="Relative path to data directory, default pcpnet")
    p.add_argument("--dataset_path", type=str, default=None, help="full path to dataset for datasets outside the local data dir")
    p.add_argument("--sparse_patches", type=int, default=False,
                   help="test on a subset of the points in each point cloud in the test data, default False")
    p.add_argument("--gpu", type=int, default=0, help="GPU to use [default: GPU 0]")
    p.add_argument("--batch_size", type=int, default=128, help="Batch size [default: 128]; the library batches internally")
    p.add_argument("--testset", type=str, default="testset_temp.txt", help="test set file name, default testset_temp.txt")
    # extensions (not in the reference)
    p.add_argument("--dtype", default="auto", choices=["auto", "f16x3c", "f16x3", "bf16x3", "f16", "bf16", "f32"],
                   help="MFMA precision mode.  auto (default): the modes that keep the output files within the reference's "
                        "tolerance (arg-max exact up to fp32 ties, 1e-5 cosine) -- f16x3c for experts_n_est (f16 hi + lo pairs "
                        "behind the two-stage gate, its margin calibrated on the first shape), f16x3 for the other models; "
                        "f16 / bf16: plain 16-bit, ~1.9x faster, hundreds of arg-max flips per 100k points (DESIGN.md 2); "
                        "f32: the exact-fp32 MFMA mode")
    p.add_argument("--subsample", default="hash", choices=["hash", "reference"],
                   help="how balls with more than num_point points are thinned: hash = on the GPU, order-independent (default); "
                        "reference = exactly like the reference (scipy cKDTree order + its RandomState stream, on the host, "
                        "~1 ms per patch) for row-by-row diffs against a real reference run")
    p.add_argument("--synthetic_weights", action="store_true", help="use seeded synthetic weights if model.nstw is absent")
    return p


def main(argv=None):
    FLAGS = build_parser().parse_args(argv)
    archs = {"experts_n_est": ARCH_EXPERTS, "ss_norm_est": ARCH_SINGLE, "ms_norm_est": ARCH_MULTI,
             "ms_sw_n_est": ARCH_SWITCH}      # test_n_est_w_experts.py / test_n_est.py / test_n_est_w_switching.py
    if FLAGS.model not in archs:
        raise SystemExit("--model must be one of %s" % sorted(archs))
    arch = archs[FLAGS.model]
    results_path = FLAGS.results_path
    base = os.getcwd()
    pc_path = FLAGS.dataset_path if FLAGS.dataset_path is not None else os.path.join(base, "data/" + FLAGS.dataset_name + "/")
    output_dir = os.path.join(results_path, FLAGS.dataset_name + "_results/")
    os.makedirs(output_dir, exist_ok=True)
    flog = open(os.path.join(output_dir, "log.txt"), "w")

    def printout(data):
        print(data)
        flog.write(data + "\n")
        sys.stdout.flush()

    model_file = os.path.join(results_path, "model.nstw")
    if os.path.exists(model_file):
        printout("Loading model %s" % model_file)
        W, cfg = wts.load(model_file)
        cfg = cfg or NestiConfig()
    elif os.path.exists(os.path.join(results_path, "model.ckpt.index")) and os.path.exists(os.path.join(results_path, "parameters.p")):
        # the reference's own artefacts: parameters.p / gmm.p / model.ckpt (test_n_est_w_experts.py:46-54, 98-105, 201)
        from . import tf_ckpt
        printout("Loading model %s" % os.path.join(results_path, "model.ckpt"))
        cfg, W = tf_ckpt.load_reference_model(results_path)
    elif FLAGS.synthetic_weights:
        cfg = NestiConfig.for_model(FLAGS.model)
        printout("No %s: using synthetic weights (seed %d)" % (model_file, wts.WEIGHT_SEED))
        W = wts.synthetic_weights(cfg)
    else:
        raise SystemExit("%s not found (pass --synthetic_weights to run without a trained model)" % model_file)
    if cfg.arch != arch:
        raise SystemExit("--model %s does not match the trained model in %s" % (FLAGS.model, results_path))
    device = "cuda:%d" % FLAGS.gpu
    dtype = FLAGS.dtype if FLAGS.dtype != "auto" else ("f16x3c" if arch == ARCH_EXPERTS else "f16x3")
    if dtype == "f16x3c" and arch != ARCH_EXPERTS:
        raise SystemExit("--dtype f16x3c is the two-stage gate of experts_n_est; use f16x3 for --model %s" % FLAGS.model)
    est = NormalEstimator(cfg, W, dtype=dtype, device=device, batch=max(FLAGS.batch_size, 4096), n_streams=2,
                          subsample=FLAGS.subsample)
    printout("Model restored.")

    dataset = PointcloudPatchDataset(pc_path, FLAGS.testset, cfg, seed=3627473, sparse_patches=FLAGS.sparse_patches,
                                     device=device)
    for ind, name in enumerate(dataset.shape_names):
        cloud = dataset.get_shape(ind)
        if dtype == "f16x3c" and ind == 0:
            # the gate margin, from up to 1024 queries of the first shape (calibrate.calibrate_gate_margin)
            from .calibrate import calibrate_gate_margin
            sp, sn = cloud.build(0, min(1024, cloud.patch_count))
            printout("gate margin tau = %.4g" % calibrate_gate_margin(est.net, sp, sn))
            del sp, sn
        normals, expert, probs = est.run(cloud)
        torch.cuda.synchronize()
        # byte-identical to the reference's np.savetxt calls (test_n_est_w_experts.py:182-188), ~6x faster
        textio.write_f32(os.path.join(output_dir, name + ".normals"), normals.cpu().numpy())
        printout("saved normals for " + name)
        if expert is None or arch == ARCH_SWITCH:   # the ablation drivers write .normals only (test_n_est.py:118,
            continue                                 # test_n_est_w_switching.py:158)
        textio.write_i32(os.path.join(output_dir, name + ".experts"), expert.cpu().numpy())
        textio.write_f32(os.path.join(output_dir, name + ".experts_probs"), probs.cpu().numpy())
        printout("saved experts for " + name)
    if dtype == "f16x3c":
        st = est.net.cascade_stats()
        printout("two-stage gate: %d of %d queries decided by the f16x3 gate, f16 gate error on a logit difference <= %.4g "
                 "(tau %.4g)" % (st["rechecked"], st["queries"], st["max_margin_err"], st["tau"]))
        if st["max_margin_err"] > 0.8 * st["tau"]:
            printout("WARNING: the f16 gate's measured error came within 20 %% of the margin on this data; an arg-max of a query "
                     "that was not rechecked may differ from the f16x3 result -- re-run with --dtype f16x3 to rule that out")
    flog.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
