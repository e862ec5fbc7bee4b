"""Normal-estimation metrics with the reference's definitions (``utils/evaluate.py:129-151,
187-198``): per-shape RMS angle (unoriented / oriented) and PGP5 / PGP10, then shape averages.
Host-side numpy; not on the hot path."""
import os

import numpy as np


def shape_metrics(normals_pred, normals_gt):
    """Both [n,3]; returns dict(rms, rms_o, pgp5, pgp10) for one shape."""
    pred = np.asarray(normals_pred, np.float32)
    gt = np.asarray(normals_gt, np.float32)
    pred = pred / np.sqrt(np.sum(np.square(pred), axis=1))[:, None]      # :131-134
    gt = gt / np.sqrt(np.sum(np.square(gt), axis=1))[:, None]
    nn = np.sum(gt * pred, axis=1)
    nn = np.clip(nn, -1, 1)                                               # :138-139
    ang = np.rad2deg(np.arccos(np.abs(nn)))                               # unoriented :141
    return {"rms": float(np.sqrt(np.mean(np.square(ang)))),               # :144
            "pgp10": float(np.sum(ang < 10.0) / float(len(ang))),         # :145
            "pgp5": float(np.sum(ang < 5.0) / float(len(ang))),           # :146
            "rms_o": float(np.sqrt(np.mean(np.square(np.rad2deg(np.arccos(nn))))))}   # :151


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
    avg = {k: float(np.mean([m[k] for m in per_shape.values()])) for k in ("rms", "rms_o", "pgp5", "pgp10")}
    return per_shape, avg
