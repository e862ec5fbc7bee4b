"""Host-side mirror of the reference's model seam.

The reference selects a model module by name and calls
``placeholder_inputs`` / ``get_model`` on it (``test_n_est_w_experts.py:60-61,74-80``,
``models/experts_n_est.py:12-108``), then executes one ``sess.run`` per batch
(``test_n_est_w_experts.py:148``).  :class:`NestiNet` plays both roles: constructing
it is "build graph + restore checkpoint", calling it is ``sess.run``.
All compute happens in ``libnesti_hip.so``.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .config import ARCH_MULTI, ARCH_SINGLE, CASCADE_DTYPES, DTYPES, NestiConfig

_TORCH_DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16,
             "bf16x3": torch.bfloat16, "f16x3": torch.float16,   # pair modes: 16-bit elements, two planes [hi | lo] per 64-channel group
             "f16x3c": torch.float16,                            # f16x3 with the two-stage gate (include/nesti_hip.h: NESTI_F16X3C)
             "f16x8": torch.float16, "f16x8c": torch.float16}    # f16x3 / f16x3c with the cross terms of the experts' tap layers at 8^3 through FP6 / FP8 (NESTI_F16X8[C])


def get_3d_grid_gmm(subdivisions=(8, 8, 8), variance=0.0156):
    """``utils/utils.py:70-95`` without sklearn: returns (weights_, means_, sigma) as
    float32 arrays, sigma = sqrt(covariances_) as fed at ``test_n_est_w_experts.py:146``."""
    n = int(subdivisions[0])
    if tuple(subdivisions) != (n, n, n):
        raise ValueError("only cubic grids are supported")
    G = n ** 3
    w = np.empty(G, np.float32)
    mu = np.empty((G, 3), np.float32)
    sg = np.empty((G, 3), np.float32)
    lib = _lib.load()
    _lib.check(lib.nesti_gmm_grid(n, float(variance), _lib.ptr(w), _lib.ptr(mu), _lib.ptr(sg)), "nesti_gmm_grid")
    return w, mu, sg


def mups_forward(cfg: NestiConfig, points, n_eff, out_dtype="f32", out_cstride=None, stream=None):
    """``get_3dmfv_n_est`` per scale + MuPS assembly (``utils/tf_util.py:655-753``,
    ``models/experts_n_est.py:66-76``).

    points [B, S*P, 3] float32 cuda, n_eff [B, S] (any integer/float dtype, like the
    uint16 placeholder ``models/experts_n_est.py:35``) -> [B, R, R, R, cstride]."""
    lib = _lib.load()
    if not points.is_cuda:
        raise _lib.NestiError("mups_forward needs CUDA/HIP tensors; there is no CPU path")
    B = points.shape[0]
    S, P, R = cfg.n_scales, cfg.num_point, cfg.n_gaussians
    if tuple(points.shape) != (B, S * P, 3):
        raise ValueError("points must be [B, %d, 3], got %s" % (S * P, tuple(points.shape)))
    points = points.contiguous().float()
    n_eff_i = n_eff.to(device=points.device, dtype=torch.int32).contiguous()
    if tuple(n_eff_i.shape) != (B, S):
        raise ValueError("n_eff must be [B, %d]" % S)
    cs = out_cstride or 20 * S
    out = torch.empty((B, R, R, R, cs), dtype=_TORCH_DT[out_dtype], device=points.device)
    c = cfg.to_c()
    st = stream if stream is not None else torch.cuda.current_stream(points.device)
    with torch.cuda.device(points.device):
        _lib.check(lib.nesti_mups_forward(ctypes.byref(c), _lib.ptr(points), _lib.ptr(n_eff_i), B, _lib.ptr(out),
                                          DTYPES[out_dtype], cs, ctypes.c_void_p(st.cuda_stream)), "nesti_mups_forward")
    return out


class NestiNet:
    """The MoE normal estimator on one GPU.

    ``weights``: dict name -> float32 ndarray in TF variable layout (see
    :mod:`.weights`).  ``dtype``: 'bf16' / 'f16' (MFMA 32x32x16, fp32 accumulate), 'f32' (exact-fp32 MFMA; the mode
    tied to the CPU oracle) or the pair modes 'f16x3' / 'bf16x3' (activations and weights as 16-bit hi + lo pairs, three
    MFMA products per multiply, a third of the 16-bit rate: f16x3 stays two orders of magnitude inside the reference's 1e-5
    cosine tolerance of the f32 mode, bf16x3 is at its edge).  'f16x3c' is f16x3 with the two-stage gate: the gating net
    runs in plain f16 first and only the queries whose f16 top-2 logit margin is below ``gate_margin`` are decided by the
    f16x3 gating net (:meth:`set_gate_margin`, :meth:`cascade_stats`).  'f16x8' / 'f16x8c' are f16x3 / f16x3c with the two cross
    terms of the experts' tap layers at 8^3 computed by one block-scaled FP6 (or FP8) MFMA (include/nesti_hip.h: NESTI_F16X8; :meth:`set_x8_layers`, :meth:`set_x8_format`)."""

    def __init__(self, cfg: NestiConfig, weights, dtype="bf16", device="cuda:0", max_batch=1024):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.NestiError("NestiNet needs a GPU: libnesti_hip.so has no CPU path")
        self.cfg, self.dtype, self.device = cfg, dtype, torch.device(device)
        self._c = cfg.to_c()
        self._handle = ctypes.c_void_p()
        names = list(weights.keys())
        self._keep = [np.ascontiguousarray(weights[k], dtype=np.float32) for k in names]
        arr = (_lib.CTensor * len(names))()
        for i, (k, a) in enumerate(zip(names, self._keep)):
            arr[i].name = k.encode()
            arr[i].data = a.ctypes.data
            arr[i].ndim = a.ndim
            for d in range(a.ndim):
                arr[i].dims[d] = a.shape[d]
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_model_create(ctypes.byref(self._c), arr, len(names), DTYPES[dtype],
                                                   ctypes.byref(self._handle)), "nesti_model_create")
        self._keep = None
        self.cascade = dtype in CASCADE_DTYPES
        self.mups_cstride = self.lib.nesti_model_mups_cstride(self._handle)
        self._ws = None
        self._ws_batch = 0
        self.reserve(max_batch)

    def __del__(self):
        try:
            h = getattr(self, "_handle", None)
            if h is not None and h.value:
                self.lib.nesti_model_destroy(h)
                h.value = None
        except Exception:      # interpreter shutdown: modules may already be torn down
            pass

    # -- two-stage gate (dtype 'f16x3c') -----------------------------------------------------
    def set_gate_margin(self, tau):
        """Queries whose f16 top-2 gate-logit margin is below ``tau`` are decided by the f16x3 gating net."""
        _lib.check(self.lib.nesti_model_set_gate_margin(self._handle, float(tau)), "nesti_model_set_gate_margin")

    def cascade_stats(self, reset=False, stream=None):
        """Counters of the two-stage gate since the last reset (synchronises the stream): dict with queries, rechecked,
        changed, max_margin_err (the f16 gate's largest error on a logit difference among the rechecked queries), sigma
        (the standard deviation of that error over all rechecked (query, expert) pairs), tau, and the self-widening
        counters: widened (queries re-decided by a widening pass), widen_events (widening passes that were not empty;
        a call runs up to ``_lib.GATE_WIDEN_PASSES``) and tau_eff = max(tau, 1.5 x max_margin_err), the threshold the next call starts from."""
        st = _lib.CCascadeStats()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_model_cascade_stats(self._handle, ctypes.byref(st), int(bool(reset)), self._stream(stream)),
                       "nesti_model_cascade_stats")
        sigma = (st.sum_sq_pair_err / st.pairs) ** 0.5 if st.pairs else 0.0
        return {"queries": int(st.queries), "rechecked": int(st.rechecked), "changed": int(st.changed),
                "max_margin_err": float(st.max_margin_err), "sigma": float(sigma), "tau": float(st.tau),
                "widened": int(st.widened), "widen_events": int(st.widen_events), "tau_eff": float(st.tau_eff)}

    def set_x8_layers(self, mask):
        """dtype 'f16x8' / 'f16x8c': which expert tap layers at 8^3 take their cross terms through the narrow format (``nesti_model_set_x8_layers``:
        bit 0 / 1 = inception1 conv2 / conv3, bit 2 / 3 = inception2 conv2 / conv3; default 0b1111; 0 = f16x3 proper)."""
        _lib.check(self.lib.nesti_model_set_x8_layers(self._handle, int(mask)), "nesti_model_set_x8_layers")

    def set_x8_format(self, bits):
        """dtype 'f16x8' / 'f16x8c': 8 = e4m3 cross terms with one scale per layer, 6 = block-scaled e2m3 (FP6: twice the FP8 matrix rate,
        ~1.2x the residual; ``nesti_model_set_x8_format``).  Recalibrate the conditioning guard afterwards."""
        _lib.check(self.lib.nesti_model_set_x8_format(self._handle, int(bits)), "nesti_model_set_x8_format")

    def set_x8_guard(self, thr):
        """dtype 'f16x8' / 'f16x8c': the conditioning guard's threshold on |n| (``nesti_model_set_x8_guard``): an expert output of
        smaller norm is re-evaluated in f16x3 proper.  ``thr < 0`` switches the guard off, ``float('inf')`` re-evaluates every query
        (what :func:`calibrate.calibrate_x8_guard` does to measure |dn|)."""
        _lib.check(self.lib.nesti_model_set_x8_guard(self._handle, float(thr)), "nesti_model_set_x8_guard")

    def x8_guard_stats(self, reset=False, stream=None):
        """Counters of the conditioning guard since the last reset (synchronises the stream): dict with queries, rechecked,
        dropped (flagged rows that did not fit an expert's list: 0 unless the threshold is absurd), max_dn (largest
        |n_x8 - n_f16x3| measured on the rows decided twice), thr and thr_eff = max(thr, 1.5 x max_dn / sqrt(2 x 2.5e-6))."""
        st = _lib.CX8GuardStats()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_model_x8_guard_stats(self._handle, ctypes.byref(st), int(bool(reset)), self._stream(stream)),
                       "nesti_model_x8_guard_stats")
        return {"queries": int(st.queries), "rechecked": int(st.rechecked), "dropped": int(st.dropped), "max_dn": float(st.max_dn),
                "thr": float(st.thr), "thr_eff": float(st.thr_eff)}

    def set_expert_mix(self, mask):
        """EXPERIMENT: which expert tap layers run a single 16-bit product (``nesti_model_set_expert_mix``; 0 = none)."""
        _lib.check(self.lib.nesti_model_set_expert_mix(self._handle, int(mask)), "nesti_model_set_expert_mix")

    def set_gate_mix(self, on):
        """EXPERIMENT: the (non-cascade) pair-mode gating net with single-product tap layers (``nesti_model_set_gate_mix``)."""
        _lib.check(self.lib.nesti_model_set_gate_mix(self._handle, int(on)), "nesti_model_set_gate_mix")

    def export_gate_error(self, dst, stream=None):
        """Write this model's ``max_margin_err`` into the one-element f32 device tensor ``dst`` (no synchronisation)."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_model_gate_error_export(self._handle, _lib.ptr(dst), self._stream(stream)),
                       "nesti_model_gate_error_export")

    def import_gate_error(self, src, stream=None):
        """Raise ``max_margin_err`` to the largest finite value of the contiguous f32 device tensor ``src`` (other ranks'
        measurements, ``dist.py``): the next call's ``tau_eff`` then covers the largest error ANY rank has seen."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_model_gate_error_import(self._handle, _lib.ptr(src), int(src.numel()), self._stream(stream)),
                       "nesti_model_gate_error_import")

    # -- workspace -------------------------------------------------------------------------
    def reserve(self, batch):
        if batch > self._ws_batch:
            nbytes = self.lib.nesti_workspace_bytes(self._handle, int(batch))
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_batch = int(batch)
        return self._ws

    def new_workspace(self, batch):
        """An additional scratch arena (one per concurrently running stream)."""
        nbytes = self.lib.nesti_workspace_bytes(self._handle, int(batch))
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def _stream(self, stream):
        """hipStream_t of ``stream`` (default: the current stream of THIS model's device)."""
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        return ctypes.c_void_p(s.cuda_stream)

    # -- pieces (tests and the reference-shaped API) ---------------------------------------
    def mups(self, points, n_eff, stream=None):
        """MuPS in the layout / dtype the towers read (``nesti_model_mups``): [B, Ri, Ri, Ri, cstride] with
        Ri = 8, or Ri = 4 for the 3^3 grid (the 27 voxels sit at [:, :3, :3, :3], the rest is zero)."""
        B = points.shape[0]
        S, P = self.cfg.n_scales, self.cfg.num_point
        if tuple(points.shape) != (B, S * P, 3):
            raise ValueError("points must be [B, %d, 3], got %s" % (S * P, tuple(points.shape)))
        points = points.contiguous().float()
        n_eff_i = n_eff.to(device=points.device, dtype=torch.int32).contiguous().view(B, S)
        Ri = round(self.lib.nesti_model_mups_rows(self._handle) ** (1.0 / 3.0))
        out = torch.empty((B, Ri, Ri, Ri, self.mups_cstride), dtype=_TORCH_DT[self.dtype], device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_model_mups(self._handle, _lib.ptr(points), _lib.ptr(n_eff_i), B, _lib.ptr(out),
                                                 self._stream(stream)), "nesti_model_mups")
        return out

    def gate(self, mups, stream=None):
        """``scale_manager_net`` + arg-max -> (probs [B,E] f32, expert [B] int32); for ms_sw_n_est
        ``noise_est_net`` + threshold -> (noise_est [B,1] f32, tower [B] int32: 0 small / 1 large)."""
        B = mups.shape[0]
        ws = self.reserve(B)
        E = self.cfg.n_gate_out
        probs = torch.empty((B, E), dtype=torch.float32, device=self.device)
        expert = torch.empty((B,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_gate_forward(self._handle, _lib.ptr(mups), B, _lib.ptr(ws), ws.numel(),
                                                   _lib.ptr(probs), _lib.ptr(expert), self._stream(stream)),
                       "nesti_gate_forward")
        return probs, expert

    def experts(self, mups, expert=None, stream=None):
        """``normal_est_net`` x E.  expert=None -> n_est [E,B,3] (reference behaviour);
        expert=[B] int32 -> top-1 routed normals [B,3]."""
        B = mups.shape[0]
        ws = self.reserve(B)
        E = self.cfg.n_towers
        if expert is None:
            out = torch.empty((E, B, 3), dtype=torch.float32, device=self.device)
            ex = None
        else:
            out = torch.zeros((B, 3), dtype=torch.float32, device=self.device)
            ex = expert.to(device=self.device, dtype=torch.int32).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_experts_forward(self._handle, _lib.ptr(mups), _lib.ptr(ex), B, _lib.ptr(ws), ws.numel(),
                                                      _lib.ptr(out), self._stream(stream)), "nesti_experts_forward")
        return out

    # -- the sess.run equivalent -----------------------------------------------------------
    def forward(self, points, n_eff, out=None, stream=None, ws=None):
        """points [B,S*P,3] f32 cuda, n_eff [B,S] -> (normals [B,3] f32, expert [B] int32, probs [B,E] f32).

        Equals ``n_est[argmax(experts_prob), range(B)]``, ``argmax`` and
        ``transpose(experts_prob)`` of ``test_n_est_w_experts.py:148-152``."""
        B = points.shape[0]
        ws = self.reserve(B) if ws is None else ws
        E = self.cfg.n_gate_out
        points = points.contiguous()
        if points.dtype != torch.float32:
            points = points.float()
        n_eff_i = n_eff if (n_eff.dtype == torch.int32 and n_eff.is_contiguous()) else n_eff.to(torch.int32).contiguous()
        if n_eff_i.dim() == 1:
            n_eff_i = n_eff_i.view(B, 1)       # ss_norm_est feeds a (B,) placeholder (models/ss_norm_est.py:30)
        if self.cfg.arch in (ARCH_SINGLE, ARCH_MULTI):      # single tower, no gate: n_pred only (test_n_est.py:136-141)
            normals = out[0] if out is not None else torch.empty((B, 3), dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(self.lib.nesti_forward(self._handle, _lib.ptr(points), _lib.ptr(n_eff_i), B, _lib.ptr(ws),
                                                  ws.numel(), _lib.ptr(normals), None, None, self._stream(stream)),
                           "nesti_forward")
            return normals, None, None
        if out is None:
            normals = torch.empty((B, 3), dtype=torch.float32, device=self.device)
            expert = torch.empty((B,), dtype=torch.int32, device=self.device)
            probs = torch.empty((B, E), dtype=torch.float32, device=self.device)
        else:
            normals, expert, probs = out
        with torch.cuda.device(self.device):
            _lib.check(self.lib.nesti_forward(self._handle, _lib.ptr(points), _lib.ptr(n_eff_i), B, _lib.ptr(ws), ws.numel(),
                                              _lib.ptr(normals), _lib.ptr(expert), _lib.ptr(probs),
                                              self._stream(stream)), "nesti_forward")
        return normals, expert, probs

    __call__ = forward


# ---- reference-shaped free functions (models/experts_n_est.py:12-108) ---------------------
def placeholder_inputs(batch_size, n_points, gmm, radius, device="cuda:0"):
    """Same tuple as ``models/experts_n_est.py:12-37``, as pre-allocated device tensors."""
    w, mu, sg = gmm
    n_rads = len(radius)
    dev = torch.device(device)
    return (torch.zeros((batch_size, n_points * n_rads, 3), dtype=torch.float32, device=dev),
            torch.zeros((batch_size, 3), dtype=torch.float32, device=dev),
            torch.as_tensor(w, device=dev), torch.as_tensor(mu, device=dev), torch.as_tensor(sg, device=dev),
            torch.zeros((batch_size, n_rads), dtype=torch.int32, device=dev))


def get_model(*args, net=None, **kwargs):
    """``models/experts_n_est.py:40-108`` return contract:
    (experts_prob [E,B], n_est [E,B,3], MuPS [B,R,R,R,20*S]).

    Two call forms.  The reference's own argument list (``models/experts_n_est.py:40``) plus ONE keyword-only argument
    carrying the restored variables -- a TF1 graph finds its weights in the session, a library call needs the handle::

        get_model(points, w, mu, sigma, is_training, radius, bn_decay=None, weight_decay=0.005,
                  original_n_points=None, n_experts=2, expert_dict=None, *, net=<NestiNet>)

    ``w`` / ``mu`` / ``sigma`` must be the Gaussian grid the model was built for (``get_3d_grid_gmm``), ``radius`` /
    ``n_experts`` / ``expert_dict`` must agree with ``net.cfg`` and ``is_training`` must be False (inference only); a
    mismatch raises instead of being ignored.  ``bn_decay`` and ``weight_decay`` only matter in training and are accepted
    for signature compatibility.  The short form ``get_model(net, points, original_n_points)`` is what the rest of this
    package uses."""
    if args and isinstance(args[0], NestiNet):
        if net is not None or kwargs or len(args) != 3:
            raise TypeError("short form: get_model(net, points, original_n_points)")
        net, points, original_n_points = args
    else:
        names = ("points", "w", "mu", "sigma", "is_training", "radius", "bn_decay", "weight_decay", "original_n_points",
                 "n_experts", "expert_dict")
        if len(args) > len(names):
            raise TypeError("get_model takes at most %d positional arguments" % len(names))
        a = dict(zip(names, args))
        for k, v in kwargs.items():
            if k not in names or k in a:
                raise TypeError("get_model: unexpected or repeated argument %r" % k)
            a[k] = v
        for k in names[:6]:
            if k not in a:
                raise TypeError("get_model: missing argument %r (models/experts_n_est.py:40)" % k)
        if net is None:
            raise TypeError("get_model: pass the restored model as net=<NestiNet> (the reference finds its variables in the session)")
        cfg = net.cfg
        if bool(a["is_training"]):
            raise ValueError("get_model: inference only (is_training must be False)")
        if len(a["radius"]) != cfg.n_scales or any(abs(float(x) - float(y)) > 1e-12 for x, y in zip(a["radius"], cfg.patch_radius)):
            raise ValueError("get_model: radius %s differs from the model's %s" % (list(a["radius"]), list(cfg.patch_radius)))
        gw, gmu, gsg = get_3d_grid_gmm((cfg.n_gaussians,) * 3, cfg.gmm_variance)
        for name, got, want in (("w", a["w"], gw), ("mu", a["mu"], gmu), ("sigma", a["sigma"], gsg)):
            got = got.detach().cpu().numpy() if hasattr(got, "detach") else np.asarray(got)
            if got.shape != want.shape or not np.allclose(got, want, rtol=0, atol=1e-6):
                raise ValueError("get_model: %s is not the %d^3 Gaussian grid (variance %g) this model was built for"
                                 % (name, cfg.n_gaussians, cfg.gmm_variance))
        if a.get("expert_dict") is not None:
            ed = {int(k): [int(x) for x in v] for k, v in dict(a["expert_dict"]).items()}
            if ed != {int(k): list(v) for k, v in cfg.expert_dict.items()}:
                raise ValueError("get_model: expert_dict differs from the model's")
            if int(a.get("n_experts", len(ed))) != cfg.n_experts:
                raise ValueError("get_model: n_experts differs from the model's %d" % cfg.n_experts)
        if a.get("original_n_points") is None:
            raise ValueError("get_model: original_n_points (the n_effective_points placeholder) is required")
        points, original_n_points = a["points"], a["original_n_points"]
    mups = net.mups(points, original_n_points)
    probs, _ = net.gate(mups)
    n_est = net.experts(mups, None)
    S, R = net.cfg.n_scales, net.cfg.n_gaussians
    return probs.t().contiguous(), n_est, mups[:, :R, :R, :R, :20 * S].float()
