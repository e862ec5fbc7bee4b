"""Normal-estimation metrics with the reference's definitions (``utils/evaluate.py:129-151,
187-198``): per-shape RMS angle (unoriented / oriented) and PGP5 / PGP10, then shape averages.
Host-side numpy; not on the hot path."""
import os

import numpy as np


def shape_metrics(normals_pred, normals_gt):
    """Both [n,3]; returns dict(rms, rms_o, pgp5, pgp10) for one shape.  The values keep the reference's scalar types
    (float32 for the angles' RMS -- the arrays are ``astype('float32')``, ``utils/evaluate.py:107-108`` -- float64 for
    the portions), so that ``str()`` of them and of their means prints what the reference's log prints."""
    pred = np.asarray(normals_pred, np.float32)
    gt = np.asarray(normals_gt, np.float32)
    pred = pred / np.sqrt(np.sum(np.square(pred), axis=1))[:, None]      # :131-134
    gt = gt / np.sqrt(np.sum(np.square(gt), axis=1))[:, None]
    nn = np.sum(gt * pred, axis=1)
    nn = np.clip(nn, -1, 1)                                               # :138-139
    ang = np.rad2deg(np.arccos(np.abs(nn)))                               # unoriented :141
    return {"rms": np.sqrt(np.mean(np.square(ang))),                      # :144
            "pgp10": np.float64(np.sum(ang < 10.0) / float(len(ang))),    # :145
            "pgp5": np.float64(np.sum(ang < 5.0) / float(len(ang))),      # :146
            "rms_o": np.sqrt(np.mean(np.square(np.rad2deg(np.arccos(nn)))))}   # :151


def evaluate_set(normal_results_path, data_path, dataset_list_file, sparse_patches=True):
    """One ``<dataset>.txt`` list (``utils/evaluate.py:52-198``): loads ``<shape>.normals`` (GT and
    predicted) and ``<shape>.pidx``; returns per-shape metrics and the shape averages."""
    with open(os.path.join(data_path, dataset_list_file)) as f:
        shapes = list(filter(None, [x.strip() for x in f.readlines()]))
    per_shape = {}
    for shape in shapes:
        gt = np.loadtxt(os.path.join(data_path, shape + ".normals")).astype("float32")
        pred = np.loadtxt(os.path.join(normal_results_path, shape + ".normals")).astype("float32")
        pidx_file = os.path.join(data_path, shape + ".pidx")
        if os.path.exists(pidx_file):
            idx = np.loadtxt(pidx_file).astype("int")
            sparse_normals = pred.shape[0] != gt.shape[0]                 # :118-122
            gt = gt[idx]
            if sparse_patches and not sparse_normals:
                pred = pred[idx]
        per_shape[shape] = shape_metrics(pred, gt)
    avg = {k: np.mean([m[k] for m in per_shape.values()]) for k in ("rms", "rms_o", "pgp5", "pgp10")}   # :187-190
    return per_shape, avg


def _list_str(values):
    """``str(list of numpy scalars)`` as the reference's interpreter prints it (numpy 1.x scalars repr as bare numbers;
    numpy 2 would print ``np.float32(...)`` wrappers)."""
    return "[" + ", ".join(str(v) for v in values) + "]"


def main(argv=None):
    """Command-line twin of the reference's ``utils/evaluate.py`` (``:22-29`` flags, ``:57-60`` summary directory,
    ``:192-198`` log lines): for each list in ``--dataset_list`` writes
    ``<normal_results_path>/summary/<dataset>_evaluation_results.txt`` and prints the same lines."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--normal_results_path", default="log/experts/pcpnet_results/", help="Log dir [default: log]")
    ap.add_argument("--data_path", type=str, default="data/pcpnet/", help="Relative path to data directory")
    ap.add_argument("--sparse_patches", type=int, default=True,
                    help="Evaluate on a sparse subset or on the entire point cloud")
    ap.add_argument("--dataset_list", type=str, default=["testset_temp"], nargs="+",
                    help="list of .txt files containing sets of point cloud names for evaluation")
    flags = ap.parse_args(argv)
    outdir = os.path.join(flags.normal_results_path, "summary/")
    os.makedirs(outdir, exist_ok=True)
    results = {}
    for dataset in flags.dataset_list:
        listfile = dataset if dataset.endswith(".txt") else dataset + ".txt"
        name = listfile[:-4]
        per_shape, avg = evaluate_set(flags.normal_results_path, flags.data_path, listfile, bool(flags.sparse_patches))
        shapes = list(per_shape)
        with open(os.path.join(outdir, name + "_evaluation_results.txt"), "w") as log:
            def log_string(out_str):
                log.write(out_str + "\n")
                print(out_str)
            log_string("RMS per shape: " + _list_str([per_shape[s]["rms"] for s in shapes]))
            log_string("RMS not oriented (shape average): " + str(avg["rms"]))
            log_string("RMS oriented (shape average): " + str(avg["rms_o"]))
            log_string("PGP10 per shape: " + _list_str([per_shape[s]["pgp10"] for s in shapes]))
            log_string("PGP5 per shape: " + _list_str([per_shape[s]["pgp5"] for s in shapes]))
            log_string("PGP10 average: " + str(avg["pgp10"]))
            log_string("PGP5 average: " + str(avg["pgp5"]))
        results[name] = avg
    return results


if __name__ == "__main__":
    main()
