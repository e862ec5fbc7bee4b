"""Opt-in compatibility path: the reference's OWN patch subsample, on the host.

The product path (``csrc/patches.hip``) keeps, when a ball holds more than P points, the P points with the smallest
hash keys -- a uniform P-subset that does not depend on batching, sharding or traversal order.  The
reference keeps ``rng.choice(n, P, replace=False)`` of scipy's cKDTree traversal-ordered ball, drawing from ONE MT19937
stream shared by every patch and scale of every shape in visiting order (``utils/pcpnet_dataset.py:237-240, 304,
320-321``).  On PCPNet's 100k clouds the largest scale overflows on every patch, so only this path can reproduce a real
reference run's ``.normals`` row for row: it restates ``__getitem__`` (``utils/pcpnet_dataset.py:286-343``, center =
'point', use_pca = False, point_tuple = 1, point_count_std = 0) with scipy and numpy on the host -- batched per library batch: ~0.2 ms per patch on 8 cores, most of it scipy's
ball query (the reference's per-patch Python loop takes 0.46 ms on one core here) -- and feeds the patch tensors to the same ``nesti_forward`` as everything else.

Pinned by ``tests/test_refsample.py``: the golden fixtures hold the reference dataset's patch tensors for queries
visited in order with its seed, capped balls included; this module reproduces them bit for bit.
"""
import itertools

import numpy as np
from scipy import spatial

REFERENCE_SEED = 3627473          # test_n_est_w_experts.py:113

# cKDTree.query_ball_point(..., return_sorted=False, workers=-1) needs scipy >= 1.6: fail at import with a clear
# message instead of a TypeError deep inside a run
import scipy  # noqa: E402
scipy_version = tuple(int(x) for x in scipy.__version__.split(".")[:2] if x.isdigit())
if scipy_version < (1, 6):
    raise ImportError("nesti_net_amd.refsample needs scipy >= 1.6 (query_ball_point return_sorted / workers), found %s" % scipy.__version__)


class RefStream:
    """The reference's shared random stream replayed natively (``csrc/refreplay.cpp``: MT19937 + numpy's legacy shuffle,
    bit-identical to ``RandomState(seed).choice(n, P, replace=False)`` ball after ball -- tests/test_refreplay.py).  Feeds the
    GPU reference-order path: ball sizes in, pick table out; only over-full balls draw (``utils/pcpnet_dataset.py:320-321``)."""

    def __init__(self, seed=REFERENCE_SEED):
        import ctypes
        from . import _lib
        self._lib, self._ct = _lib, ctypes
        self.lib = _lib.load()
        self._h = ctypes.c_void_p(self.lib.nesti_refstream_create(int(seed) & 0xffffffff))
        if not self._h.value:
            raise _lib.NestiError("nesti_refstream_create failed")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.lib.nesti_refstream_destroy(self._h)
                self._h.value = None
        except Exception:      # interpreter shutdown
            pass

    def picks(self, sizes, P, out=None):
        """``sizes``: int32 ball sizes in visiting order (patch-major, scale-minor).  Returns (picks uint16 [n_over * P],
        offsets int64 [len(sizes)], -1 where the ball holds <= P points).  ``out``: optional (picks, offsets) buffers to fill
        (picks at least ``count(sizes > P) * P`` long), e.g. pinned host tensors' numpy views."""
        sizes = np.ascontiguousarray(sizes, dtype=np.int32).ravel()
        n_over = int((sizes > P).sum())
        if out is None:
            picks, offsets = np.empty(n_over * P, np.uint16), np.empty(len(sizes), np.int64)
        else:
            picks, offsets = out
            if picks.dtype != np.uint16 or offsets.dtype != np.int64 or len(offsets) < len(sizes) or not picks.flags.c_contiguous:
                raise ValueError("out = (uint16 picks, int64 offsets[len(sizes)])")
        got = self._ct.c_int64(0)
        self._lib.check(self.lib.nesti_refstream_picks(self._h, self._lib.ptr(sizes), len(sizes), int(P), self._lib.ptr(picks), picks.size,
                                                       self._lib.ptr(offsets), self._ct.byref(got)), "nesti_refstream_picks")
        return picks[:got.value * P], offsets[:len(sizes)]


class ReferencePatchSampler:
    """Holds the random stream of one reference ``PointcloudPatchDataset`` (``utils/pcpnet_dataset.py:237-240``).
    Patches must be requested in the reference's visiting order (shapes in list order, patch rows in order,
    ``SequentialPointcloudPatchSampler``) for the stream to line up with a reference run."""

    def __init__(self, seed=REFERENCE_SEED, stream=None):
        """``stream``: a :class:`RefStream` to draw the picks from instead of a numpy ``RandomState`` -- the same numbers
        (tests/test_refreplay.py) without one Python call per over-full ball, and ONE stream object that the GPU reference-order
        path (pipeline.py) can share, so that a shape may fall back to this host path without losing the stream position."""
        self.seed = int(seed)
        self.stream = stream
        self.rng = np.random.RandomState(self.seed) if stream is None else None
        self._bufs = {}            # (M, S, P) -> (points, idx): reused from batch to batch (first-touch page faults of a fresh
                                   # 180 MB tensor cost as much as the arithmetic that fills it)

    @staticmethod
    def build_tree(pts):
        return spatial.cKDTree(pts, 10)                                     # utils/pcpnet_dataset.py:37

    def patches(self, pts, tree, center_inds, r_abs, P):
        """``pts`` [N,3] float32, ``center_inds`` [M] point indices (the shape's .pidx rows or a range), ``r_abs`` the
        absolute radii (Python floats, :282) -> points [M, S*P, 3] float32, n_eff [M, S] int32.  ``points`` is one of two
        buffers the sampler alternates between: it stays valid until the call AFTER the next one.

        Batched: ONE ``query_ball_point`` per scale for all M centres (all cores; ``return_sorted=False`` keeps cKDTree's
        traversal order, which is what a single-point query returns and what the reference subsamples), then the shared
        random stream is replayed in the reference's visiting order -- patch-major, scale-minor, one
        ``choice(n, P, replace=False)`` per ball that holds more than P points (the number of MT19937 draws of a choice
        depends on its rejections, so the stream itself stays sequential) -- and the gather / centre / scale arithmetic runs
        vectorised per scale.  A 100k-point cloud takes seconds instead of the ~100 s of one Python call per patch and scale."""
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        center_inds = np.asarray(center_inds, dtype=np.int64)
        M, S = len(center_inds), len(r_abs)
        n_eff = np.zeros((M, S), np.int32)
        if M == 0:
            return np.zeros((0, S * P, 3), np.float32), n_eff
        # two alternating buffer sets: the batch handed out by the PREVIOUS call stays valid while this one is being filled
        # (NormalEstimator overlaps the host pass of batch k + 1 with the upload + GPU forward of batch k)
        self._flip = 1 - getattr(self, "_flip", 1)
        key = (M, S, P, self._flip)
        if key not in self._bufs:
            self._bufs = {k: v for k, v in self._bufs.items() if k[:3] == (M, S, P)}
            self._bufs[key] = (np.empty((M, S * P, 3), np.float32), np.empty((M, P), np.int32))
        points, idx = self._bufs[key]            # every element is overwritten below (:298's zero rows included)
        centers = pts[center_inds]
        balls = [tree.query_ball_point(centers, rad, return_sorted=False, workers=-1) for rad in r_abs]     # :304, all centres
        sizes = np.stack([np.fromiter(map(len, balls[s]), dtype=np.int64, count=M) for s in range(S)], axis=1)
        n_eff[:] = np.minimum(sizes, P)                                     # :310-311
        # the random stream, in visiting order; only over-full balls draw from it (:320-321)
        over = sizes > P
        picks = [None] * S
        pick_rows = [np.nonzero(over[:, s])[0] for s in range(S)]
        if self.stream is not None:                                         # native replay: row-major = patch-major, scale-minor
            flat_picks, offs = self.stream.picks(sizes.astype(np.int32).ravel(), P)
            offs = offs.reshape(M, S)
            table = flat_picks.reshape(-1, P)
            for s in range(S):
                picks[s] = table[offs[pick_rows[s], s] // P].astype(np.int64)
        else:
            choice = self.rng.choice
            for s in range(S):
                picks[s] = np.empty((len(pick_rows[s]), P), np.int64)
            fill = [0] * S
            for i, s in zip(*np.nonzero(over)):                             # row-major = patch-major, scale-minor
                picks[s][fill[s]] = choice(int(sizes[i, s]), P, replace=False)
                fill[s] += 1
        for s, rad in enumerate(r_abs):
            # all balls of the scale flattened in one pass (row order, traversal order within a row)
            flat = np.fromiter(itertools.chain.from_iterable(balls[s]), dtype=np.int64, count=int(sizes[:, s].sum()))
            balls[s] = None                                                 # the Python lists of this scale are no longer needed
            off = np.cumsum(sizes[:, s]) - sizes[:, s]
            # neighbour indices row by row; a padding slot points at the centre itself: (c - c) / r = +0.0, the zero row of :298
            idx[:] = center_inds[:, None]
            small = np.nonzero(~over[:, s])[0]
            cnt = sizes[small, s]
            rows = np.repeat(small, cnt)
            within = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            idx[rows, within] = flat[np.repeat(off[small], cnt) + within]
            if len(pick_rows[s]):
                idx[pick_rows[s]] = flat[off[pick_rows[s]][:, None] + picks[s]]
            # :330-343: float32 gather, minus the centre, divided by the radius as a float32 scalar; rows beyond n_eff stay zero
            # (ndarray.take on the flattened index list is ~10x faster than fancy-indexing with the 2-D index array)
            dst = points[:, s * P:(s + 1) * P]
            np.subtract(pts.take(idx.ravel(), axis=0).reshape(M, P, 3), centers[:, None, :], out=dst)
            np.divide(dst, np.float32(rad), out=dst)
        return points, n_eff
