"""Host-side mirror of the reference's data seam.

``provider.get_data_loader(...)`` (``utils/provider.py:319-429``) returns
``(DataLoader, PointcloudPatchDataset)`` whose batches are
``([b,S*P,3] f32, [b,3,3], [b,S] f64)``.  :func:`get_data_loader` here has the same
keyword arguments for the inference configuration (``test_n_est_w_experts.py:109-116``)
and yields the same tuple, but the patches are produced by the HIP ball-query kernel
(``nesti_patches_build``) from a cloud resident in HBM instead of one Python
``__getitem__`` per point.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .config import NestiConfig


def load_xyz(path):
    """``np.loadtxt(point_filename).astype('float32')`` with the ``.npy`` cache the reference
    writes next to the file (``utils/pcpnet_dataset.py:249-251``, ``:13-14``)."""
    npy = path + ".npy"
    if os.path.exists(npy) and os.path.getmtime(npy) >= os.path.getmtime(path):
        return np.load(npy)
    pts = np.loadtxt(path).astype("float32")
    if pts.ndim == 1:
        pts = pts.reshape(1, -1)
    pts = np.ascontiguousarray(pts[:, :3])
    try:
        np.save(npy, pts)
    except OSError:
        pass   # read-only dataset dir: the cache is an optimisation only
    return pts


class CloudPatches:
    """One shape: the cloud in HBM + its uniform search grid (replaces ``load_shape`` /
    ``cKDTree``, ``utils/pcpnet_dataset.py:13-39``).  ``build(rows)`` extracts patches."""

    def __init__(self, pts, cfg: NestiConfig, device="cuda:0", seed=3627473, pidx=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.NestiError("CloudPatches needs a GPU: libnesti_hip.so has no CPU path")
        self.cfg, self.device, self.seed = cfg, torch.device(device), int(seed)
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        self.host_pts = pts                   # kept for the opt-in reference-order subsample (pipeline.py: subsample='reference')
        self.n_points = pts.shape[0]
        # utils/pcpnet_dataset.py:281-282 -- float64 Python arithmetic on the host, like the reference
        self.bbdiag = float(np.linalg.norm(pts.max(0) - pts.min(0), 2))
        self.r_abs = [self.bbdiag * rad for rad in cfg.patch_radius]
        self.cloud = torch.from_numpy(pts).to(self.device)
        if pidx is not None:
            pidx = np.asarray(pidx).astype(np.int64).reshape(-1)
            if len(pidx) and (pidx.min() < 0 or pidx.max() >= self.n_points):      # a bad .pidx file must not read out of bounds
                raise ValueError("pidx entries must lie in [0, %d): got [%d, %d]" % (self.n_points, pidx.min(), pidx.max()))
        self.pidx = None if pidx is None else torch.as_tensor(pidx, dtype=torch.int32, device=self.device)
        self.patch_count = self.n_points if pidx is None else len(pidx)     # :276-279
        nbytes = self.lib.nesti_patches_workspace_bytes(self.n_points)
        self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._c = cfg.to_c()
        self._r = (ctypes.c_double * len(self.r_abs))(*self.r_abs)
        self.build_grid()

    def build_grid(self, stream=None):
        """(Re)build the uniform search grid on the device -- the cKDTree construction of the
        reference (``utils/pcpnet_dataset.py:37``).  Asynchronous on ``stream``."""
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_patches_grid(ctypes.byref(self._c), _lib.ptr(self.cloud), self.n_points, self._r,
                                                   _lib.ptr(self._ws), self._ws.numel(), ctypes.c_void_p(st.cuda_stream)),
                       "nesti_patches_grid")

    def build(self, first, count, want_idx=False, out=None, stream=None):
        """Patches for local patch rows [first, first+count) (file order, or ``pidx`` order when
        sparse -- ``utils/pcpnet_dataset.py:292-295``).

        Returns points [count,S*P,3] f32, n_eff [count,S] int32 (+ nbr_idx [count,S*P], n_ball
        [count,S] when ``want_idx``).  The subsample key uses the global row ``first + i`` so
        results do not depend on batching."""
        S, P = self.cfg.n_scales, self.cfg.num_point
        if first < 0 or count < 0 or first + count > self.patch_count:
            raise ValueError("patch rows [%d, %d) outside [0, %d)" % (first, first + count, self.patch_count))
        # sparse: <shape>.pidx rows; full: NULL -> the kernel uses point index = patch row
        qidx = self.pidx[first:first + count].contiguous() if self.pidx is not None else None
        if out is None:
            points = torch.empty((count, S * P, 3), dtype=torch.float32, device=self.device)
            n_eff = torch.empty((count, S), dtype=torch.int32, device=self.device)
        else:
            points, n_eff = out
        nbr = torch.empty((count, S * P), dtype=torch.int32, device=self.device) if want_idx else None
        n_ball = torch.empty((count, S), dtype=torch.int32, device=self.device) if want_idx else None
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_patches_query(
                ctypes.byref(self._c), _lib.ptr(self.cloud), self.n_points, _lib.ptr(qidx), count, self._r,
                ctypes.c_uint64(self.seed), ctypes.c_int(first), _lib.ptr(points), _lib.ptr(n_eff), _lib.ptr(nbr),
                _lib.ptr(n_ball), _lib.ptr(self._ws), self._ws.numel(), ctypes.c_void_p(st.cuda_stream)),
                "nesti_patches_query")
        if want_idx:
            return points, n_eff, nbr, n_ball
        return points, n_eff

    # ---- the reference's own subsample order on the GPU (utils/pcpnet_dataset.py:304, 320-321; patches.hip: patches_ref_kernel) ----
    def ensure_tree_order(self):
        """cKDTree's visiting order as a sort key: ``scipy.spatial.cKDTree(pts, 10)`` exactly as the reference builds it
        (``utils/pcpnet_dataset.py:37``); ``query_ball_point`` returns a ball in ascending position in ``tree.indices``
        (tests/test_refreplay.py), so the kernel only needs ``rank[i]`` = that position and ``order`` = ``tree.indices``."""
        if getattr(self, "_tree_rank", None) is None:
            from scipy import spatial
            tree = spatial.cKDTree(self.host_pts, 10)
            order = np.ascontiguousarray(tree.indices, dtype=np.int32)
            rank = np.empty(self.n_points, np.int32)
            rank[order] = np.arange(self.n_points, dtype=np.int32)
            self._ref_tree = tree            # the host path (refsample.ReferencePatchSampler) uses the same tree
            self._tree_order = torch.from_numpy(order).to(self.device)
            self._tree_rank = torch.from_numpy(rank).to(self.device)
        return self._tree_rank, self._tree_order

    def count_balls(self, first, count, stream=None):
        """Ball sizes [count, S] int32 (device) of patch rows [first, first + count): what the reference's random stream needs
        (``nesti_patches_count``)."""
        S = self.cfg.n_scales
        if first < 0 or count < 0 or first + count > self.patch_count:
            raise ValueError("patch rows [%d, %d) outside [0, %d)" % (first, first + count, self.patch_count))
        qidx = self.pidx[first:first + count].contiguous() if self.pidx is not None else None
        n_ball = torch.empty((count, S), dtype=torch.int32, device=self.device)
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_patches_count(ctypes.byref(self._c), _lib.ptr(self.cloud), self.n_points, _lib.ptr(qidx), count,
                                                    self._r, ctypes.c_int(first), _lib.ptr(n_ball), _lib.ptr(self._ws), self._ws.numel(),
                                                    ctypes.c_void_p(st.cuda_stream)), "nesti_patches_count")
        return n_ball

    def build_reference_order(self, first, count, picks, pick_offsets, want_idx=False, out=None, stream=None):
        """Patch tensors of rows [first, first + count) exactly as the reference's ``PointcloudPatchDataset.__getitem__`` builds
        them (``nesti_patches_query_ref``): balls in cKDTree's visiting order, over-full balls thinned by ``picks`` (uint16 device
        tensor, any 2-byte dtype) at ``pick_offsets`` [count * S] int64 (device; -1 = the ball holds <= P points) -- the table
        ``refsample.RefStream.picks`` replays from the ball sizes of :meth:`count_balls`."""
        S, P = self.cfg.n_scales, self.cfg.num_point
        if first < 0 or count < 0 or first + count > self.patch_count:
            raise ValueError("patch rows [%d, %d) outside [0, %d)" % (first, first + count, self.patch_count))
        rank, order = self.ensure_tree_order()
        qidx = self.pidx[first:first + count].contiguous() if self.pidx is not None else None
        if out is None:
            points = torch.empty((count, S * P, 3), dtype=torch.float32, device=self.device)
            n_eff = torch.empty((count, S), dtype=torch.int32, device=self.device)
        else:
            points, n_eff = out
        nbr = torch.empty((count, S * P), dtype=torch.int32, device=self.device) if want_idx else None
        if pick_offsets.dtype != torch.int64 or pick_offsets.numel() < count * S or picks.element_size() != 2:
            raise ValueError("picks: 2-byte elements, pick_offsets: int64 [count * S]")
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_patches_query_ref(
                ctypes.byref(self._c), _lib.ptr(self.cloud), self.n_points, _lib.ptr(qidx), count, self._r, ctypes.c_int(first),
                _lib.ptr(rank), _lib.ptr(order), _lib.ptr(picks) if picks.numel() else None, _lib.ptr(pick_offsets), _lib.ptr(points),
                _lib.ptr(n_eff), _lib.ptr(nbr), _lib.ptr(self._ws), self._ws.numel(), ctypes.c_void_p(st.cuda_stream)),
                "nesti_patches_query_ref")
        if want_idx:
            return points, n_eff, nbr
        return points, n_eff


class PointcloudPatchDataset:
    """Shape list + per-shape patch counts, like the reference class of the same name
    (``utils/pcpnet_dataset.py:179-282``), for the inference configuration only."""

    def __init__(self, root, shape_list_filename, cfg: NestiConfig, seed=3627473, sparse_patches=False,
                 device="cuda:0", cache_capacity=100):
        self.root, self.cfg, self.seed, self.device = root, cfg, seed, device
        self.sparse_patches = bool(sparse_patches)
        with open(os.path.join(root, shape_list_filename)) as f:
            self.shape_names = [x.strip() for x in f.readlines()]
        self.shape_names = list(filter(None, self.shape_names))       # :221-224
        self.shape_patch_count = []
        self._cache, self._cache_cap, self._tick = {}, cache_capacity, 0
        for ind in range(len(self.shape_names)):
            self.shape_patch_count.append(self.get_shape(ind).patch_count)

    def get_shape(self, ind):
        """LRU of loaded shapes (``Cache``, ``utils/pcpnet_dataset.py:151-176``)."""
        self._tick += 1
        if ind not in self._cache:
            if len(self._cache) >= self._cache_cap:
                old = min(self._cache, key=lambda k: self._cache[k][1])
                del self._cache[old]
            name = self.shape_names[ind]
            pts = load_xyz(os.path.join(self.root, name + ".xyz"))
            pidx = None
            if self.sparse_patches:
                pidx = np.loadtxt(os.path.join(self.root, name + ".pidx")).astype("int")   # :267-270
            self._cache[ind] = [CloudPatches(pts, self.cfg, self.device, self.seed, pidx), self._tick]
        self._cache[ind][1] = self._tick
        return self._cache[ind][0]

    def __len__(self):
        return sum(self.shape_patch_count)


class _Loader:
    """Iterates all patches of all shapes in order (``SequentialPointcloudPatchSampler``,
    ``utils/pcpnet_dataset.py:41-55``) in batches of ``batch_size``; a batch may span shapes."""

    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size = dataset, int(batch_size)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        ds, B = self.dataset, self.batch_size
        pend_p, pend_n, have = [], [], 0
        for ind in range(len(ds.shape_names)):
            shape, done = ds.get_shape(ind), 0
            while done < shape.patch_count:
                take = min(B - have, shape.patch_count - done)
                p, n = shape.build(done, take)
                pend_p.append(p)
                pend_n.append(n)
                have += take
                done += take
                if have == B:
                    yield self._emit(pend_p, pend_n)
                    pend_p, pend_n, have = [], [], 0
        if have:
            yield self._emit(pend_p, pend_n)

    @staticmethod
    def _emit(ps, ns):
        p = ps[0] if len(ps) == 1 else torch.cat(ps)
        n = ns[0] if len(ns) == 1 else torch.cat(ns)
        trans = torch.eye(3, device=p.device).expand(p.shape[0], 3, 3)     # use_pca=False (:377)
        return p, trans, n


def get_data_loader(dataset_name, batchSize, indir, patch_radius, points_per_patch, outputs=(), patch_point_count_std=0,
                    seed=3627473, identical_epochs=False, use_pca=False, patch_center="point", point_tuple=1,
                    cache_capacity=100, patch_sample_order="full", workers=0, dataset_type="test", sparse_patches=False,
                    cfg: NestiConfig = None, device="cuda:0"):
    """Keyword-compatible with ``utils/provider.py:319-429`` for the inference settings of
    ``test_n_est_w_experts.py:109-116``; anything outside that configuration raises, as it is
    outside the hot path."""
    if outputs not in ((), [], None) or use_pca or patch_center != "point" or point_tuple != 1 \
            or patch_point_count_std != 0 or identical_epochs or patch_sample_order != "full":
        raise ValueError("only the inference configuration of test_n_est_w_experts.py:109-116 is supported")
    cfg = cfg or NestiConfig()
    cfg = NestiConfig(**{**cfg.__dict__, "patch_radius": list(patch_radius), "num_point": int(points_per_patch)})
    listfile = os.path.relpath(dataset_name, indir) if os.path.isabs(dataset_name) else dataset_name
    ds = PointcloudPatchDataset(indir, listfile, cfg, seed=seed, sparse_patches=sparse_patches, device=device,
                                cache_capacity=cache_capacity)
    return _Loader(ds, batchSize), ds
