"""Synthetic-weights helper: make the gating net spread its arg-max over all experts.

With random weights the gate's output is almost input-independent, so every query routes to one
expert; a trained gate (no checkpoint ships with the reference) does not behave like that.  One probe
forward over a sample of real MuPS tensors measures the pre-softmax logits of the last gating layer
(``fc4noise``, ``models/experts_n_est.py:172-177``); its weights are then rescaled per expert so the
logits have equal spread around a common level and every expert wins about 1/E of the sample.
Only the synthetic ``fc4noise`` variables change; the graph and every kernel are untouched."""
import numpy as np
import torch

from .model import NestiNet

BASE = 5.0      # common logit level, far enough above 0 that fc4's ReLU never clips


def calibrate_gate(cfg, W, points, n_eff, device="cuda:0", spread=2.0):
    """Return a copy of ``W`` whose gate routes the sample ``points``/``n_eff`` (device tensors,
    [B,S*P,3] / [B,S]) roughly uniformly."""
    E = cfg.n_experts
    Wp = dict(W)
    w4 = np.array(W["fc4noise/weights"], dtype=np.float32, copy=True)       # [128, E]
    b4 = np.full((E,), BASE, np.float32)
    w4[:, 0] = 0.0                                                          # expert 0: constant logit BASE
    w4 *= 0.05 / max(1e-6, float(np.abs(w4).max()))                         # small: no ReLU clipping in the probe
    Wp["fc4noise/weights"], Wp["fc4noise/biases"] = w4, b4
    probe = NestiNet(cfg, Wp, dtype="f32", device=device, max_batch=int(points.shape[0]))
    mups = probe.mups(points, n_eff)
    probs, _ = probe.gate(mups)
    torch.cuda.synchronize()
    del probe
    p = probs.double().cpu().numpy()
    d = np.log(p[:, 1:] / p[:, :1])                                         # = w_e.h (bias terms cancel), e >= 1
    mean, std = d.mean(0), np.maximum(d.std(0), 1e-9)
    g = spread / std
    w4n = w4.copy()
    w4n[:, 1:] = w4[:, 1:] * g[None, :].astype(np.float32)
    b4n = b4.copy()
    b4n[1:] = (BASE - mean * g).astype(np.float32)                          # logit_e = BASE + spread * z_e
    z = BASE + (d - mean) * g                                               # calibrated logits of experts >= 1
    b4n[0] = np.float32(np.quantile(z.max(1), 1.0 / E))                     # expert 0 wins ~1/E of the sample
    out = dict(W)
    out["fc4noise/weights"], out["fc4noise/biases"] = w4n, b4n
    return out


GATE_MARGIN_SIGMAS = 7.0     # tau = this many standard deviations of the f16 gate's error on a logit difference
GATE_MARGIN_OVER_MAX = 1.5   # ... and at least this x the largest such error seen on the calibration sample: the library's own
                             # widening factor (NESTI_GATE_WIDEN), so that a freshly calibrated margin starts un-widened


GATE_MARGIN_MIN_QUERIES = 256   # fewer calibration queries than this: no filtering (every query is decided by the f16x3 gate)


def calibrate_gate_margin(net, points, n_eff, sigmas=GATE_MARGIN_SIGMAS, over_max=GATE_MARGIN_OVER_MAX, floor=1e-3):
    """Set the gate margin tau of an 'f16x3c' :class:`NestiNet` (``nesti_model_set_gate_margin``) from a sample of queries
    (``points`` [B,S*P,3] / ``n_eff`` [B,S] device tensors).

    With the margin at infinity every query of the sample goes through both gating passes, which measures the plain-f16
    pass's error on the logit differences against its own arg-max -- the only way it can flip an arg-max.  That error is
    rounding noise: zero-mean, independent of the margin itself and close to Gaussian (on the bench's 100k cloud sigma =
    0.021, kurtosis 3.4, the largest of 600 000 pair errors 0.124 = 5.9 sigma), so tau = max(``sigmas`` x sigma,
    ``over_max`` x the sample's largest error).  The guarantee is statistical, and it is enforced rather than assumed: every
    later forward call re-measures the error on the queries it decides twice (an unbiased sample, because the error does
    not depend on the margin) and the LIBRARY raises its threshold to 1.5 x the largest error seen so far, re-deciding the
    rows in between in the same call (include/nesti_hip.h: NESTI_GATE_WIDEN; :meth:`NestiNet.cascade_stats` reports
    ``widened`` / ``widen_events`` / ``tau_eff``).  The counters are reset here, so the threshold starts at tau.  Returns tau.
    A sample of fewer than ``GATE_MARGIN_MIN_QUERIES`` queries cannot carry a 7-sigma statement: the margin is then left at
    infinity, which makes the mode plain f16x3 (always safe, no filter gain)."""
    if int(points.shape[0]) < GATE_MARGIN_MIN_QUERIES:
        net.cascade_stats(reset=True)         # the counters start from zero for this shape here too
        net.set_gate_margin(1e30)
        return float("inf")
    net.cascade_stats(reset=True)
    net.set_gate_margin(1e30)
    net.gate(net.mups(points, n_eff))
    st = net.cascade_stats(reset=True)
    tau = max(float(sigmas) * st["sigma"], float(over_max) * st["max_margin_err"], float(floor))
    net.set_gate_margin(tau)
    return tau


X8_GUARD_MIN_QUERIES = 256


def calibrate_x8_guard(net, points, n_eff, floor=0.05):
    """Set the conditioning guard's threshold of an 'f16x8' / 'f16x8c' :class:`NestiNet` (``nesti_model_set_x8_guard``) from a sample
    of queries (``points`` [B,S*P,3] / ``n_eff`` [B,S] device tensors).

    With the threshold at infinity every query of the sample is evaluated by its expert twice -- FP8 cross terms, then f16x3
    proper -- which measures |dn| = |n_x8 - n_f16x3|, the only way the FP8 layers can tilt a normal.  |dn| does not depend on |n|
    (corr 0.1 on the bench cloud), so the rows a later call re-evaluates keep measuring it without bias, and the LIBRARY raises its
    threshold to ``_lib.X8_GUARD_WIDEN`` x the largest |dn| seen / sqrt(2 x ``_lib.X8_GUARD_BAR``) by itself.  The threshold set here
    is that formula on the sample's largest |dn| (at least ``floor``); the counters are reset, so it starts un-widened.  A sample
    of fewer than ``X8_GUARD_MIN_QUERIES`` queries leaves the library default in place.  The gate margin of an 'f16x8c' model is
    calibrated separately (:func:`calibrate_gate_margin`); call this one afterwards -- it runs a full forward pass.  Returns thr."""
    from . import _lib
    if int(points.shape[0]) < X8_GUARD_MIN_QUERIES:
        net.x8_guard_stats(reset=True)
        net.set_x8_guard(_lib.X8_GUARD_DEFAULT)
        return _lib.X8_GUARD_DEFAULT
    net.x8_guard_stats(reset=True)
    net.set_x8_guard(float("inf"))
    net.forward(points, n_eff)
    st = net.x8_guard_stats(reset=True)
    thr = max(float(floor), _lib.X8_GUARD_WIDEN * st["max_dn"] / (2.0 * _lib.X8_GUARD_BAR) ** 0.5)
    net.set_x8_guard(thr)
    return thr
