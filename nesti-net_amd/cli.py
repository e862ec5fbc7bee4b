"""Command-line drop-in for the reference's ``test_n_est_w_experts.py``: same flags
(``:19-29``), same inputs (``<dataset_path>/<testset>`` shape list, ``<shape>.xyz``, optional
``<shape>.pidx``) and same outputs (``<results_path>/<dataset_name>_results/<shape>.normals``,
``.experts``, ``.experts_probs`` written with ``np.savetxt`` like ``:182-188``, plus ``log.txt``).

The trained-model directory holds either ``model.nstw`` (variables + hyper-parameters, see
:mod:`.weights`) or the reference's own ``parameters.p`` / ``gmm.p`` / ``model.ckpt.*`` (read by
:mod:`.tf_ckpt` without TensorFlow); ``--synthetic_weights`` substitutes seeded random weights (no
checkpoint ships with the reference)."""
import argparse
import ctypes
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
    p.add_argument("--dtype", default="auto", choices=["auto", "f16x8c", "f16x8", "f16x3c", "f16x3", "bf16x3", "f16", "bf16", "f32"],
                   help="MFMA precision mode.  auto (default): the modes that keep .normals and .experts within the reference's "
                        "tolerance (arg-max exact up to fp32 ties, 1e-5 cosine) -- f16x8c for experts_n_est on the 8^3 grid (f16 hi + lo "
                        "pairs behind the two-stage gate, whose margin is calibrated on every shape and widens itself when the measured "
                        "error approaches it; the experts' tap layers at 8^3 take their two cross terms through one block-scaled FP6 MFMA and outputs of "
                        "small norm are evaluated again in f16x3: .experts equal f16x3c's, .normals stay within ~1e-6 cosine of f16x3's), f16x3c on the 3^3 grid, f16x3 for the other "
                        "models.  NOTE: with f16x3c / f16x8c the .experts_probs rows of queries "
                        "the filter pass decided alone (~90 %%) are that pass's probabilities, within ~0.013 of the fp32 values; "
                        "use --dtype f16x3 when the probabilities themselves must hold 1e-4.  f16 / bf16: plain 16-bit, ~1.7x "
                        "faster, hundreds of arg-max flips per 100k points (DESIGN.md 2); f32: the exact-fp32 MFMA mode")
    p.add_argument("--x8_format", type=int, default=None, choices=[6, 8],
                   help="dtypes f16x8 / f16x8c: the experts' cross terms as block-scaled FP6 e2m3 (6, the default: half the matrix-pipe time of "
                        "FP8 for ~1.15x its residual, bounded by the same conditioning guard) or FP8 e4m3 (8)")
    p.add_argument("--x8_layers", type=int, default=None,
                   help="dtypes f16x8 / f16x8c: which expert tap layers at 8^3 take their cross terms through the narrow format (bit 0 / 1 = inception1 "
                        "conv2 (3^3) / conv3 (5^3), bit 2 / 3 = inception2 conv2 / conv3).  Default 15 = all four (outputs of small norm are "
                        "re-evaluated in f16x3 by the conditioning guard: 1 - cos <= 1.1e-6 against f16x3 on every other query); 10 = the "
                        "5^3 layers only (~2 %% slower); 0 = f16x3c proper")
    p.add_argument("--lib_batch", type=int, default=0,
                   help="queries per library batch (0 = by dtype: 50000 for f16x3c / f16 / bf16, 25000 for the pair modes, 8192 "
                        "for f32; two such batches are in flight on two HIP streams)")
    p.add_argument("--subsample", default="hash", choices=["hash", "reference", "reference_host"],
                   help="how balls with more than num_point points are thinned: hash = on the GPU, order-independent (default); "
                        "reference = exactly like the reference (cKDTree visiting order + its shared RandomState stream: "
                        "utils/pcpnet_dataset.py:304, 320-321) for row-by-row diffs against a real reference run -- on the GPU since "
                        "round 6 (ball sizes counted on the device, the stream replayed natively on the host from the sizes, the balls "
                        "sorted into tree order in LDS); reference_host = the same rows by scipy + numpy on the host "
                        "(nesti-net_amd/refsample.py), several times slower")
    p.add_argument("--synthetic_weights", action="store_true", help="use seeded synthetic weights if model.nstw is absent")
    return p


def fit_batch(cfg, dtype, batch, device, lanes=2, reserve=2 << 30):
    """The largest batch <= ``batch`` (halving, 256-row granularity, >= 256) whose ``lanes`` arenas fit the device's free
    memory with ``reserve`` bytes to spare.  The arena size comes from the library's own layout, from the configuration alone
    (``nesti_estimate_workspace_bytes_for_config`` == ``nesti_estimate_workspace_bytes`` of the model that will be created:
    tests/test_abi.py), before any model exists."""
    from . import _lib
    from .config import DTYPES
    lib = _lib.load()
    free = torch.cuda.mem_get_info(device)[0] - reserve
    c = cfg.to_c()
    while batch > 256:
        arena = lib.nesti_estimate_workspace_bytes_for_config(ctypes.byref(c), DTYPES[dtype], batch)
        if arena and lanes * arena <= free:
            break
        batch = max(256, (batch // 2 + 255) // 256 * 256)
    return batch


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
    from .config import CASCADE_DTYPES
    dtype = FLAGS.dtype if FLAGS.dtype != "auto" else (("f16x8c" if cfg.n_gaussians == 8 else "f16x3c") if arch == ARCH_EXPERTS else "f16x3")
    if dtype in CASCADE_DTYPES + ("f16x8",) and arch != ARCH_EXPERTS:
        raise SystemExit("--dtype %s belongs to experts_n_est (two-stage gate / narrow-format cross terms in the expert towers); use f16x3 for "
                         "--model %s" % (dtype, FLAGS.model))
    if dtype in ("f16x8", "f16x8c") and cfg.n_gaussians != 8:
        raise SystemExit("--dtype %s needs the 8^3 Gaussian grid; use f16x3c" % dtype)
    dataset = PointcloudPatchDataset(pc_path, FLAGS.testset, cfg, seed=3627473, sparse_patches=FLAGS.sparse_patches,
                                     device=device)
    # two library batches in flight on two HIP streams; a batch is half the largest shape (rounded up to 256 rows) unless
    # that exceeds what the workspace of the dtype allows (~2 MB per query in f16x3c, twice that in the full pair modes)
    lib_batch = FLAGS.lib_batch or {"f16x3c": 50000, "f16x8c": 50000, "f16": 50000, "bf16": 50000, "f32": 8192}.get(dtype, 25000)
    half = (max(dataset.shape_patch_count + [1]) + 1) // 2
    batch = max(FLAGS.batch_size, min(lib_batch, max(1024, (half + 255) // 256 * 256)))
    # ... and what the device has free right now: two arenas (one per stream) + the 1024-query calibration workspace must fit
    # (the defaults are sized for an otherwise idle 288 GB MI355X; a shared or smaller device gets smaller batches, not an OOM)
    fit = fit_batch(cfg, dtype, batch, device, lanes=2)
    if fit != batch:
        printout("library batch %d -> %d rows: %.1f GB free on %s" % (batch, fit, torch.cuda.mem_get_info(device)[0] / 1e9, device))
        batch = fit
    if (FLAGS.x8_layers is not None or FLAGS.x8_format is not None) and dtype not in ("f16x8", "f16x8c"):
        raise SystemExit("--x8_layers / --x8_format belong to --dtype f16x8 / f16x8c")
    est = NormalEstimator(cfg, W, dtype=dtype, device=device, batch=batch, n_streams=2, subsample=FLAGS.subsample,
                          x8_layers=FLAGS.x8_layers, x8_format=FLAGS.x8_format)
    printout("Model restored.")

    for ind, name in enumerate(dataset.shape_names):
        cloud = dataset.get_shape(ind)
        if dtype in CASCADE_DTYPES:
            # the gate margin, from up to 1024 queries of THIS shape (noise level and density change the activation and
            # error statistics from shape to shape); calibrate_gate_margin resets the gate's counters, so the statistics
            # printed below are this shape's
            from .calibrate import calibrate_gate_margin
            sp, sn = cloud.build(0, min(1024, cloud.patch_count))
            printout("gate margin for %s: tau = %.4g" % (name, calibrate_gate_margin(est.net, sp, sn)))
            del sp, sn
        if dtype in ("f16x8", "f16x8c"):
            # the conditioning guard of the FP6 / FP8 cross-term layers: its |n| threshold from the same sample of THIS shape
            from .calibrate import calibrate_x8_guard
            sp, sn = cloud.build(0, min(1024, cloud.patch_count))
            printout("cross-term guard threshold for %s: |n| < %.4g" % (name, calibrate_x8_guard(est.net, sp, sn)))
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
        if dtype in CASCADE_DTYPES:
            st = est.net.cascade_stats()
            printout("two-stage gate on %s: %d of %d queries decided by the f16x3 gate, f16 gate error on a logit difference "
                     "<= %.4g (tau %.4g, threshold now %.4g)" % (name, st["rechecked"], st["queries"], st["max_margin_err"],
                                                                st["tau"], st["tau_eff"]))
            if st["widen_events"]:
                printout("  the measured error came within a factor 1.5 of the margin: the library widened it and re-decided "
                         "%d more queries with the f16x3 gate before these files were written" % st["widened"])
        if dtype in ("f16x8", "f16x8c"):
            gs = est.net.x8_guard_stats()
            printout("narrow-format cross terms on %s: %d of %d expert outputs re-evaluated in f16x3 (|n| below %.4g), largest |dn| measured %.3g"
                     % (name, gs["rechecked"], gs["queries"], gs["thr_eff"], gs["max_dn"]))
            if gs["dropped"]:
                printout("  WARNING: %d flagged outputs did not fit the guard's lists and keep their narrow-format cross-term values: the "
                         "1 - cos <= 2.5e-6 bound against f16x3 is not established for them (use --dtype f16x3c for this shape)" % gs["dropped"])
    flog.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
