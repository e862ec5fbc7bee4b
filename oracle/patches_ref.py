"""ORACLE (test infrastructure, NOT product code): CPU restatement of the reference's
multi-scale patch extraction, ``utils/pcpnet_dataset.py:286-343`` (``__getitem__`` with
center='point', use_pca=False, point_tuple=1, point_count_std=0).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.

Parity status: **pinned** against golden patches captured from the reference's own
``PointcloudPatchDataset`` in this container (``scripts/make_golden_patches.py`` ->
``tests/golden/patches_*.npz``): ball sets, ``n_eff``, and -- wherever the ball holds at most
P points -- the patch rows themselves, bit for bit.

One documented deviation: when a ball holds more than P points the reference keeps
``rng.choice(n, P, replace=False)`` of cKDTree's traversal-ordered list with one MT19937
stream shared by all patches (``:320-321``), which cannot be reproduced without scipy's tree.
Both this oracle and the HIP kernel keep instead the P points with the smallest
``(subsample_hash(seed, query_row, scale, index), index)`` keys, rows in key order -- also a
uniform P-subset, but a pure function of its inputs.
"""
import numpy as np
from scipy import spatial

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def subsample_hash(seed, q, s, idx):
    """splitmix64 finaliser; must match ``subsample_hash`` in csrc/patches.hip."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) ^ (np.uint64(q) << np.uint64(34)) ^ (np.uint64(s) << np.uint64(32)) ^ idx)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(32)).astype(np.uint32)


def patch_radii(pts, patch_radius):
    """``utils/pcpnet_dataset.py:281-282``: bbdiag and absolute radii as Python floats."""
    bbdiag = float(np.linalg.norm(pts.max(0) - pts.min(0), 2))
    return bbdiag, [bbdiag * rad for rad in patch_radius]


def build_tree(pts):
    return spatial.cKDTree(pts, 10)      # utils/pcpnet_dataset.py:37


def extract_patches(pts, query_idx, r_abs, P, seed, tree=None, rows=None):
    """pts [N,3] float32; query_idx [M]; r_abs list of S python floats; ``rows`` [M]: the patch-row number of each query
    within its shape, which keys the documented subsample (default: its position in ``query_idx``, i.e. a shape whose
    patch rows are exactly these queries in this order).

    Returns points [M,S*P,3] f32, n_eff [M,S] i32, nbr [M,S*P] i32 (-1 padded),
    n_ball [M,S] i32 (uncapped ball sizes)."""
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    tree = tree or build_tree(pts)
    M, S = len(query_idx), len(r_abs)
    points = np.zeros((M, S * P, 3), np.float32)                     # :298
    n_eff = np.zeros((M, S), np.int32)
    nbr = np.full((M, S * P), -1, np.int32)
    n_ball = np.zeros((M, S), np.int32)
    for q, c in enumerate(query_idx):
        center = pts[c, :]
        for s, rad in enumerate(r_abs):
            inds = np.array(tree.query_ball_point(center, rad), dtype=np.int64)   # :304
            n_ball[q, s] = len(inds)
            count = min(P, len(inds))                                # :310
            n_eff[q, s] = count
            h = subsample_hash(seed, q if rows is None else int(rows[q]), s, inds)
            order = np.lexsort((inds, h))[:count]                    # documented subsample rule
            inds = inds[order]
            start = s * P
            nbr[q, start:start + count] = inds
            # (pts[inds] - pts[c]) / rad, all in float32 (:330-343; torch f32 tensor / python float)
            points[q, start:start + count] = (pts[inds] - center) / np.float32(rad)
    return points, n_eff, nbr, n_ball
