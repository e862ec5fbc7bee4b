"""Opt-in compatibility path: the reference's OWN patch subsample, on the host.

The product path (``csrc/patches.hip``) keeps, when a ball holds more than P points, the P points with the smallest
hash keys -- a uniform P-subset that does not depend on batching, sharding or traversal order.  The
reference keeps ``rng.choice(n, P, replace=False)`` of scipy's cKDTree traversal-ordered ball, drawing from ONE MT19937
stream shared by every patch and scale of every shape in visiting order (``utils/pcpnet_dataset.py:237-240, 304,
320-321``).  On PCPNet's 100k clouds the largest scale overflows on every patch, so only this path can reproduce a real
reference run's ``.normals`` row for row: it restates ``__getitem__`` (``utils/pcpnet_dataset.py:286-343``, center =
'point', use_pca = False, point_tuple = 1, point_count_std = 0) with scipy and numpy on the host -- about a millisecond
per patch, like the reference -- and feeds the patch tensors to the same ``nesti_forward`` as everything else.

Pinned by ``tests/test_refsample.py``: the golden fixtures hold the reference dataset's patch tensors for queries
visited in order with its seed, capped balls included; this module reproduces them bit for bit.
"""
import numpy as np
from scipy import spatial

REFERENCE_SEED = 3627473          # test_n_est_w_experts.py:113


class ReferencePatchSampler:
    """Holds the random stream of one reference ``PointcloudPatchDataset`` (``utils/pcpnet_dataset.py:237-240``).
    Patches must be requested in the reference's visiting order (shapes in list order, patch rows in order,
    ``SequentialPointcloudPatchSampler``) for the stream to line up with a reference run."""

    def __init__(self, seed=REFERENCE_SEED):
        self.seed = int(seed)
        self.rng = np.random.RandomState(self.seed)

    @staticmethod
    def build_tree(pts):
        return spatial.cKDTree(pts, 10)                                     # utils/pcpnet_dataset.py:37

    def patches(self, pts, tree, center_inds, r_abs, P):
        """``pts`` [N,3] float32, ``center_inds`` [M] point indices (the shape's .pidx rows or a range), ``r_abs`` the
        absolute radii (Python floats, :282) -> points [M, S*P, 3] float32, n_eff [M, S] int32."""
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        M, S = len(center_inds), len(r_abs)
        points = np.zeros((M, S * P, 3), np.float32)                        # :298 (.zero_())
        n_eff = np.zeros((M, S), np.int32)
        for i, c in enumerate(center_inds):
            center = pts[c, :]
            for s, rad in enumerate(r_abs):
                inds = np.array(tree.query_ball_point(center, rad))         # :304
                count = min(P, len(inds))                                   # :310
                n_eff[i, s] = count
                if count < len(inds):                                       # :320-321
                    inds = inds[self.rng.choice(len(inds), count, replace=False)]
                # :330-343: float32 gather, minus the centre, divided by the radius as a float32 scalar
                points[i, s * P:s * P + count] = (pts[inds.astype(np.int64)] - center) / np.float32(rad)
        return points, n_eff
