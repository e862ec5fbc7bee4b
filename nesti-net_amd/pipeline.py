"""End-to-end hot path for one shape: cloud (host or device) -> multi-scale patches (HIP ball
query) -> MuPS -> gating -> top-1 expert -> normals.  This is the body of the reference's
``predict`` loop (``test_n_est_w_experts.py:129-152``) with the per-point Python
``__getitem__`` and the TF ``sess.run`` replaced by three library calls per batch."""
import ctypes

import numpy as np
import torch

from . import _lib
from .config import ARCH_MULTI, ARCH_SINGLE, NestiConfig
from .model import NestiNet
from .provider import CloudPatches


class NormalEstimator:
    """``use_graph=True`` captures the forward pass of a full batch (MuPS + gate + routing + experts,
    ~250 launches whose sizes depend only on the batch size -- expert counts are read on the device) into a
    hipGraph once and replays it; only worthwhile for small batches where launch gaps matter."""

    def __init__(self, cfg: NestiConfig, weights, dtype="bf16", device="cuda:0", batch=4096, seed=3627473,
                 use_graph=False, n_streams=1, gate_margin=None, subsample="hash", x8_layers=None, x8_format=None):
        self.cfg, self.device, self.batch, self.seed = cfg, torch.device(device), int(batch), seed
        # subsample='reference': balls larger than P are thinned exactly like the reference does (scipy cKDTree traversal
        # order + ONE numpy RandomState stream over all patches in visiting order, utils/pcpnet_dataset.py:304-321), on the
        # host, ~0.5 ms per patch -- for diffing against a real reference run row by row (refsample.py).  The default 'hash'
        # is the GPU ball query with its order-independent uniform subset.
        # Since round 6 'reference' runs on the GPU: ball sizes from a count pass, the random stream replayed natively on the host
        # from those sizes (csrc/refreplay.cpp, a worker thread), the balls sorted into cKDTree's visiting order by the patch
        # kernel (patches.hip: patches_ref_kernel) -- row for row what 'reference_host' (the scipy + numpy path) produces, at
        # several times its speed.  A shape with a ball too large for the kernel's LDS sort falls back to the host path; both
        # draw from ONE native stream object, so the stream position survives the switch.
        if subsample not in ("hash", "reference", "reference_host"):
            raise ValueError("subsample must be 'hash', 'reference' or 'reference_host'")
        self.subsample = subsample
        self._ref = None
        self._ref_failed = False
        if subsample != "hash":
            from .refsample import ReferencePatchSampler, RefStream
            self._ref_stream = RefStream(seed)
            self._ref = ReferencePatchSampler(seed, stream=self._ref_stream)
        self.use_graph = bool(use_graph)
        # n_streams > 1: consecutive batches alternate between HIP streams (own scratch arena each), so one batch's partially
        # filled last workgroup rounds overlap the other's kernels
        self.n_streams = 1 if (use_graph or self._ref is not None) else max(1, int(n_streams))
        # fused mode (everything but the hipGraph and reference-subsample modes): ONE library call per batch stream
        # (nesti_estimate_normals / _multi: fused ball query + MuPS kernel, gate, routing, experts) on an arena that holds the
        # staging buffers and the forward workspace; with one stream that is one call per run
        self._fused = not self.use_graph and self._ref is None
        self.net = NestiNet(cfg, weights, dtype=dtype, device=device, max_batch=1 if self._fused else self.batch)
        if gate_margin is not None:                       # dtypes 'f16x3c' / 'f16x8c' only (calibrate.calibrate_gate_margin picks one)
            self.net.set_gate_margin(gate_margin)
        if x8_layers is not None:                         # dtypes 'f16x8' / 'f16x8c' only (NestiNet.set_x8_layers; default 0b1111)
            self.net.set_x8_layers(x8_layers)
        if x8_format is not None:                         # ... and in which format (NestiNet.set_x8_format: 6 = block-scaled e2m3, the default; 8 = e4m3)
            self.net.set_x8_format(x8_format)
        S, P, E = cfg.n_scales, cfg.num_point, max(1, cfg.n_gate_out)
        self._graph = None
        if self._fused:
            nbytes = self.net.lib.nesti_estimate_workspace_bytes(self.net._handle, self.batch)
            self._arena = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            # lanes 1 .. n_streams - 1: (stream, arena); lane 0 is the caller's current stream with self._arena
            self._lanes = [(torch.cuda.Stream(device=self.device), torch.empty(nbytes, dtype=torch.uint8, device=self.device))
                           for _ in range(1, self.n_streams)]
            return
        self._lanes = []
        self._points = torch.empty((self.batch, S * P, 3), dtype=torch.float32, device=self.device)
        self._n_eff = torch.empty((self.batch, S), dtype=torch.int32, device=self.device)
        if self.use_graph:
            self._g_out = (torch.empty((self.batch, 3), dtype=torch.float32, device=self.device),
                           torch.empty((self.batch,), dtype=torch.int32, device=self.device),
                           torch.empty((self.batch, E), dtype=torch.float32, device=self.device))

    def _capture(self):
        """Warm up (sets kernel attributes, fills the workspace pointers) then capture one forward."""
        self._points.zero_()
        self._n_eff.fill_(1)
        with torch.cuda.device(self.device):
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.net.forward(self._points, self._n_eff, out=self._g_out)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(self.device)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self.net.forward(self._points, self._n_eff, out=self._g_out)

    def prepare(self, pts, pidx=None):
        """Upload a cloud and build its search grid (replaces ``load_shape``)."""
        return CloudPatches(pts, self.cfg, device=self.device, seed=self.seed, pidx=pidx)

    def run(self, cloud: CloudPatches, first=0, count=None, out=None):
        """Normals for patch rows [first, first+count) of a prepared cloud.

        Returns (normals [count,3] f32, expert [count] int32, probs [count,E] f32) on the device
        (ms_sw_n_est: expert = tower picked by the noise threshold, probs = noise_est [count,1]);
        everything is enqueued on the current stream, nothing synchronises."""
        count = cloud.patch_count - first if count is None else count
        E = max(1, self.cfg.n_gate_out)
        if out is None:
            normals = torch.empty((count, 3), dtype=torch.float32, device=self.device)
            expert = torch.empty((count,), dtype=torch.int32, device=self.device)
            probs = torch.empty((count, E), dtype=torch.float32, device=self.device)
        else:
            normals, expert, probs = out
        single_tower = self.cfg.arch in (ARCH_SINGLE, ARCH_MULTI)      # ss/ms ablations: normals only
        if first < 0 or count < 0 or first + count > cloud.patch_count:
            raise ValueError("patch rows [%d, %d) outside [0, %d)" % (first, first + count, cloud.patch_count))
        if self._ref is not None:
            if self._ref_failed:
                raise _lib.NestiError("a reference-order run failed half way: the shared random stream no longer lines up with "
                                      "the reference's; create a new NormalEstimator")
            if self.subsample == "reference" and count > 0:
                return self._run_reference_gpu(cloud, first, count, normals, expert, probs, single_tower)
            return self._run_reference_order(cloud, first, count, normals, expert, probs, single_tower)
        if self._fused:
            main = torch.cuda.current_stream(self.device)
            for st, _ in self._lanes:
                st.wait_stream(main)                  # the cloud's grid (and earlier work) is ready
            # one stream: ONE call, the library walks the batches; several: one call per batch, alternating the lanes
            step = count if self.n_streams == 1 else self.batch
            done, it = 0, 0
            while done < count:
                take = min(step, count - done)
                lane = it % self.n_streams
                it += 1
                st, arena = (main, self._arena) if lane == 0 else self._lanes[lane - 1]
                r0 = first + done
                qidx = cloud.pidx[r0:r0 + take].contiguous() if cloud.pidx is not None else None
                with torch.cuda.device(self.device):
                    _lib.check(self.net.lib.nesti_estimate_normals(
                        self.net._handle, _lib.ptr(cloud.cloud), cloud.n_points, _lib.ptr(qidx), take, cloud._r,
                        ctypes.c_uint64(cloud.seed), r0, self.batch, 0, _lib.ptr(cloud._ws), cloud._ws.numel(),
                        _lib.ptr(arena), arena.numel(), _lib.ptr(normals[done:done + take]),
                        None if single_tower else _lib.ptr(expert[done:done + take]),
                        None if single_tower else _lib.ptr(probs[done:done + take]),
                        ctypes.c_void_p(st.cuda_stream)), "nesti_estimate_normals")
                if qidx is not None and lane > 0:
                    qidx.record_stream(st)
                done += take
            for st, _ in self._lanes:
                main.wait_stream(st)                  # results are ordered on the caller's stream again
            if single_tower:
                return normals, None, None
            return normals, expert, probs
        done = 0
        while done < count:
            take = min(self.batch, count - done)
            sl = slice(done, done + take)
            p, n = self._points[:take], self._n_eff[:take]
            cloud.build(first + done, take, out=(p, n))
            if self.use_graph and take == self.batch:
                if self._graph is None:
                    self._capture()
                    cloud.build(first + done, take, out=(p, n))     # capture ran on placeholder inputs
                self._graph.replay()
                normals[sl].copy_(self._g_out[0])
                expert[sl].copy_(self._g_out[1])
                probs[sl].copy_(self._g_out[2])
            else:
                self.net.forward(p, n, out=(normals[sl], expert[sl], probs[sl]))
            done += take
        if single_tower:
            return normals, None, None
        return normals, expert, probs

    def _prefetch(self, spans, make, consume):
        """Producer / consumer over ``spans`` with two buffers in flight: ``make(i, span)`` runs in ONE worker thread, span after
        span (the shared random stream stays in visiting order), ``consume(i, span, item)`` on the calling thread; the worker
        may run one span ahead of the span being consumed.  Whatever fails, the worker is stopped and joined before the error
        propagates, and the estimator refuses further reference-order runs (the stream position is lost)."""
        import queue
        import threading
        ready, free, stop = queue.Queue(), threading.Semaphore(2), threading.Event()

        def produce():
            try:
                for i, span in enumerate(spans):
                    free.acquire()
                    if stop.is_set():
                        return
                    ready.put(make(i, span))
            except BaseException as e:      # noqa: BLE001 -- hand the failure to the consumer instead of dying silently
                ready.put(e)

        worker = threading.Thread(target=produce, daemon=True)
        worker.start()
        ok = False
        try:
            for i, span in enumerate(spans):
                item = ready.get()
                if isinstance(item, BaseException):
                    raise item
                consume(i, span, item, free.release)
            ok = True
        finally:
            if not ok:
                self._ref_failed = True
                stop.set()
                free.release()
                free.release()
            worker.join()

    def _run_reference_order(self, cloud, first, count, normals, expert, probs, single_tower):
        """Patch rows [first, first + count) with the reference's own subsample on the HOST (scipy ball query + the replayed random
        stream, ``refsample.py``), then the usual forward pass.  The host pass of batch k + 1 runs in a worker thread while batch k is
        uploaded and goes through the GPU."""
        if getattr(cloud, "_ref_tree", None) is None:
            cloud._ref_tree = self._ref.build_tree(cloud.host_pts)
        S, P = self.cfg.n_scales, self.cfg.num_point
        pidx_host = cloud.pidx[first:first + count].cpu().numpy() if cloud.pidx is not None else None
        spans = [(done, min(self.batch, count - done)) for done in range(0, count, self.batch)]

        def make(i, span):
            done, take = span
            centers = pidx_host[done:done + take] if pidx_host is not None else np.arange(first + done, first + done + take)
            return self._ref.patches(cloud.host_pts, cloud._ref_tree, centers, cloud.r_abs, P)

        def consume(i, span, item, release):
            done, take = span
            p, n = item
            sl = slice(done, done + take)
            p_d, n_d = torch.from_numpy(p).to(self.device), torch.from_numpy(n).to(self.device).view(take, S)   # synchronous copies
            release()                           # the host buffer may be refilled
            self.net.forward(p_d, n_d, out=(normals[sl], expert[sl], probs[sl]))

        self._prefetch(spans, make, consume)
        if single_tower:
            return normals, None, None
        return normals, expert, probs

    def _run_reference_gpu(self, cloud, first, count, normals, expert, probs, single_tower):
        """The same rows on the GPU.  cKDTree's result order is ascending position in ``tree.indices``, so
        the reference's subsample needs (1) the ball sizes -- one count launch for the whole range, (2) its random stream replayed
        over those sizes in visiting order -- natively, in a worker thread, one batch ahead (``refsample.RefStream``), (3) the
        balls sorted by tree position with the picks applied -- ``patches_ref_kernel``, then the usual forward pass.  The patch
        tensors equal the host path's bit for bit (tests/test_gpu_patches.py), hence so do the normals."""
        S, P = self.cfg.n_scales, self.cfg.num_point
        cloud.ensure_tree_order()
        sizes = cloud.count_balls(first, count).cpu().numpy()                  # [count, S]; synchronises
        cap = min(int(self.net.lib.nesti_patches_ref_max_ball()), 65535)
        if int(sizes.max(initial=0)) > cap:
            # a ball too large for the kernel's LDS sort (or the uint16 pick table): this shape goes through the host path, which
            # draws from the same stream object
            return self._run_reference_order(cloud, first, count, normals, expert, probs, single_tower)
        spans = [(done, min(self.batch, count - done)) for done in range(0, count, self.batch)]
        if getattr(self, "_ref_pinned", None) is None:
            n = self.batch * S * P
            self._ref_pinned = [(torch.empty(n, dtype=torch.int16).pin_memory(), torch.empty(self.batch * S, dtype=torch.int64).pin_memory())
                                for _ in range(2)]
            self._ref_dev = [(torch.empty(n, dtype=torch.int16, device=self.device),
                              torch.empty(self.batch * S, dtype=torch.int64, device=self.device)) for _ in range(2)]
            self._ref_copy = torch.cuda.Stream(device=self.device)

        def make(i, span):
            done, take = span
            pk, off = self._ref_pinned[i & 1]
            picks, _ = self._ref_stream.picks(sizes[done:done + take].ravel(), P, out=(pk.numpy().view(np.uint16), off.numpy()))
            return len(picks)

        main = torch.cuda.current_stream(self.device)
        copied = [None, None]

        def consume(i, span, n_picks, release):
            done, take = span
            pk, off = self._ref_pinned[i & 1]
            pk_d, off_d = self._ref_dev[i & 1]
            if copied[i & 1] is not None:
                self._ref_copy.wait_event(copied[i & 1])       # the kernel that read this device slot two batches ago is done
            with torch.cuda.stream(self._ref_copy):
                pk_d[:n_picks].copy_(pk[:n_picks], non_blocking=True)
                off_d[:take * S].copy_(off[:take * S], non_blocking=True)
            self._ref_copy.synchronize()        # only the copies: the compute stream keeps running
            release()                           # the pinned slot may be refilled
            main.wait_stream(self._ref_copy)
            sl = slice(done, done + take)
            p, n = self._points[:take], self._n_eff[:take]
            cloud.build_reference_order(first + done, take, pk_d, off_d, out=(p, n))
            ev = torch.cuda.Event()
            ev.record(main)
            copied[i & 1] = ev
            self.net.forward(p, n, out=(normals[sl], expert[sl], probs[sl]))

        self._prefetch(spans, make, consume)
        if single_tower:
            return normals, None, None
        return normals, expert, probs

    def run_many(self, items):
        """Several shapes (or shards of shapes) as ONE stream of batches: ``items`` = [(cloud, first, count), ...].
        Returns one (normals, expert, probs) triple of device tensors per item (views of the concatenated outputs).
        Small shapes / shards share the gate and expert launches (``nesti_estimate_normals_multi``); results equal
        ``run`` on each item.  With ``n_streams`` > 1 the stream of rows is cut into groups of ``batch`` rows and the
        groups alternate between the HIP streams.  8^3 grid, fused mode; otherwise falls back to per-item ``run``."""
        if not self._fused or self.cfg.n_gaussians != 8:
            return [self.run(c, f, n) for c, f, n in items]
        total = sum(n for _, _, n in items)
        E = max(1, self.cfg.n_gate_out)
        single_tower = self.cfg.arch in (ARCH_SINGLE, ARCH_MULTI)
        normals = torch.empty((total, 3), dtype=torch.float32, device=self.device)
        expert = torch.empty((total,), dtype=torch.int32, device=self.device)
        probs = torch.empty((total, E), dtype=torch.float32, device=self.device)
        for cloud, first, count in items:
            if first < 0 or count < 0 or first + count > cloud.patch_count:
                raise ValueError("patch rows [%d, %d) outside [0, %d)" % (first, first + count, cloud.patch_count))
        # groups of pieces (cloud, first, count): one group = one library call on one lane
        if self.n_streams == 1:
            groups = [list(items)]
        else:
            groups, cur, room = [], [], self.batch
            for cloud, first, count in items:
                while count > 0:
                    take = min(count, room)
                    cur.append((cloud, first, take))
                    first, count, room = first + take, count - take, room - take
                    if room == 0:
                        groups.append(cur)
                        cur, room = [], self.batch
            if cur:
                groups.append(cur)
        main = torch.cuda.current_stream(self.device)
        for st, _ in self._lanes:
            st.wait_stream(main)
        keep, o = [], 0
        for gi, group in enumerate(groups):
            lane = gi % self.n_streams
            st, arena = (main, self._arena) if lane == 0 else self._lanes[lane - 1]
            arr = (_lib.CShapeQueries * max(1, len(group)))()
            for i, (cloud, first, count) in enumerate(group):
                qidx = cloud.pidx[first:first + count].contiguous() if cloud.pidx is not None else None
                if qidx is not None and lane > 0:
                    qidx.record_stream(st)
                keep.append(qidx)
                arr[i].cloud_dev = cloud.cloud.data_ptr()
                arr[i].n_points = cloud.n_points
                arr[i].query_idx_dev = qidx.data_ptr() if qidx is not None and count > 0 else None
                arr[i].n_queries = count
                for s, r in enumerate(cloud.r_abs):
                    arr[i].r_abs[s] = r
                arr[i].seed = cloud.seed
                arr[i].query_row0 = first
                arr[i].grid_ws_dev = cloud._ws.data_ptr()
                arr[i].grid_ws_bytes = cloud._ws.numel()
            rows = sum(c for _, _, c in group)
            with torch.cuda.device(self.device):
                _lib.check(self.net.lib.nesti_estimate_normals_multi(
                    self.net._handle, arr, len(group), self.batch, _lib.ptr(arena), arena.numel(), _lib.ptr(normals[o:o + rows]),
                    None if single_tower else _lib.ptr(expert[o:o + rows]), None if single_tower else _lib.ptr(probs[o:o + rows]),
                    ctypes.c_void_p(st.cuda_stream)), "nesti_estimate_normals_multi")
            o += rows
        for st, _ in self._lanes:
            main.wait_stream(st)
        out, o = [], 0
        for _, _, n in items:
            out.append((normals[o:o + n], None if single_tower else expert[o:o + n], None if single_tower else probs[o:o + n]))
            o += n
        return out

    def estimate(self, pts, pidx=None):
        """Convenience: numpy cloud in, numpy results out (synchronises)."""
        cloud = self.prepare(np.asarray(pts, dtype=np.float32), pidx)
        normals, expert, probs = self.run(cloud)
        torch.cuda.synchronize(self.device)
        if expert is None:
            return normals.cpu().numpy(), None, None
        return normals.cpu().numpy(), expert.cpu().numpy(), probs.cpu().numpy()
