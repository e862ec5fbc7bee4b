"""Parity statistics between two runs of the hot path on the same queries (no oracle involved).

The north star asks for the expert arg-max bit-exact and the normals within 1e-5 cosine of the
reference's fp32 CPU result (``test_n_est_w_experts.py:150-152``, ``models/experts_n_est.py:174-177``).
The exact-fp32 MFMA mode (``dtype='f32'``) is the mode tied to the fp64 CPU oracle (tests/test_gpu_net.py,
tests/test_gpu_fixtures.py: ~1.8k queries over six fixture clouds); a faster mode is characterised against it on the
full workload: every arg-max difference is counted, and ``meets_north_star`` is the strict reading of the clause --

* no arg-max difference at all, except where the fp32 reference's OWN top-2 probabilities are closer than
  ``TIE_MARGIN``: an fp32 evaluation in another summation order (the reference's Eigen convolutions, this library's f32
  MFMA mode, or the fp64 oracle) does not define the arg-max of such a query, so it cannot be "bit-exact" against
  anything; these are reported as ``argmax_ties``.  ``TIE_MARGIN`` is not hand-picked: it is 2 x ``F32_PROB_ERR_BOUND``,
  the largest difference between the f32 mode's and the fp64 oracle's probabilities over every query of the six fixture
  clouds with a calibrated gate (each of the top two probabilities can move by that much, so a gap below twice the
  bound can close); tests/test_gpu_fixtures.py measures the difference on every run, prints it and fails if it exceeds
  the bound, so the constant below is pinned by the suite (measured: see ``F32_PROB_ERR_MEASURED``);
* 1 - cos <= 1e-5 on EVERY query whose arg-max agrees (a tie that resolved the other way returns another expert's normal
  and is reported through ``max_incl_flips``).

``strict_bit_exact_argmax`` is the even stricter boolean (zero differences, ties included).
"""
import numpy as np

COS_TOL = 1e-5            # north star: cosine tolerance on the normal vectors
# |p_f32mode - p_fp64oracle| over all queries of the six fixture clouds (tests/test_gpu_fixtures.py asserts <= the bound on
# every fixture and prints the measured value; the arithmetic is deterministic, so the figure is the same on every box)
F32_PROB_ERR_MEASURED = 6.0e-5
F32_PROB_ERR_BOUND = 6.5e-5
TIE_MARGIN = 2 * F32_PROB_ERR_BOUND   # the fp32 reference's own top-2 probabilities closer than this: fp32 arithmetic does not define the arg-max


def _cos(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    den = np.maximum(np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1), 1e-300)
    return (a * b).sum(-1) / den


def compare(test, ref, tie_margin=TIE_MARGIN, cos_tol=COS_TOL):
    """``test`` / ``ref``: (normals [n,3], expert [n] or None, probs [n,E] or None) as numpy arrays, ``ref`` from the
    exact-fp32 mode.  Returns a JSON-serialisable dict."""
    n_t, e_t, p_t = test
    n_r, e_r, p_r = ref
    n = len(n_r)
    out = {"queries": int(n), "reference": "same library, exact-fp32 MFMA mode (dtype f32), same queries",
           "oracle_queries": "the f32 mode itself is held to the fp64 CPU oracle on ~1.8k queries of six fixture clouds "
                             "(tests/test_gpu_fixtures.py); all weights are synthetic (no checkpoint ships with the reference)",
           "cos_tol": cos_tol}
    omc = 1.0 - _cos(n_t, n_r)
    if e_r is None:                       # single-tower models: no gate
        same = np.ones(n, bool)
    else:
        same = np.asarray(e_t) == np.asarray(e_r)
        srt = np.sort(np.asarray(p_r, np.float64), axis=1)
        margin = srt[:, -1] - srt[:, -2] if srt.shape[1] > 1 else np.full(n, np.inf)
        flips = ~same
        out.update({
            "argmax_match_rate": float(same.mean()) if n else 1.0,
            "argmax_flips": int(flips.sum()),
            "tie_margin": tie_margin,
            "argmax_ties": int((flips & (margin < tie_margin)).sum()),
            "flips_outside_margin": int((flips & (margin >= tie_margin)).sum()),
            "flip_margin_max": float(margin[flips].max()) if flips.any() else 0.0,
            "queries_within_margin": int((margin < tie_margin).sum()),
            "prob_abs_err_max": float(np.abs(np.asarray(p_t, np.float64) - np.asarray(p_r, np.float64)).max()) if n else 0.0,
        })
    m = omc[same]
    q = (lambda v, x: float(np.quantile(v, x))) if len(m) else (lambda v, x: 0.0)
    out["one_minus_cos"] = {"p50": q(m, 0.5), "p99": q(m, 0.99), "max": float(m.max()) if len(m) else 0.0,
                            "over": "queries whose arg-max agrees (a flipped query is a different expert's normal)",
                            "max_incl_flips": float(omc.max()) if n else 0.0}
    out["meets_north_star"] = bool(out.get("flips_outside_margin", 0) == 0 and out["one_minus_cos"]["max"] <= cos_tol)
    out["strict_bit_exact_argmax"] = bool(out.get("argmax_flips", 0) == 0)
    return out
