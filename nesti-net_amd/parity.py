"""Parity statistics between two runs of the hot path on the same queries (no oracle involved).

The north star asks for the expert arg-max bit-exact and the normals within 1e-5 cosine of the
reference's fp32 CPU result (``test_n_est_w_experts.py:150-152``, ``models/experts_n_est.py:174-177``).
The exact-fp32 MFMA mode (``dtype='f32'``) is the mode tied to the CPU oracle (tests/test_gpu_net.py,
tests/test_gpu_fixtures.py); a production dtype (f16 / bf16) is characterised against it on the full
workload: arg-max match rate, every flip counted and classified by the reference's top-2 probability
margin, and the distribution of 1 - cos between the normal vectors.
"""
import numpy as np

COS_TOL = 1e-5            # north star: cosine tolerance on the normal vectors
MARGIN_FLAG = 2e-3        # an arg-max flip is "margin-flagged" when the fp32 run's top-2 probabilities are closer than this


def _cos(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    den = np.maximum(np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1), 1e-300)
    return (a * b).sum(-1) / den


def compare(test, ref, margin_flag=MARGIN_FLAG, cos_tol=COS_TOL):
    """``test`` / ``ref``: (normals [n,3], expert [n] or None, probs [n,E] or None) as numpy arrays, ``ref`` from the
    exact-fp32 mode.  Returns a JSON-serialisable dict."""
    n_t, e_t, p_t = test
    n_r, e_r, p_r = ref
    n = len(n_r)
    out = {"queries": int(n), "reference": "same library, exact-fp32 MFMA mode (dtype f32), same queries",
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
            "margin_flag": margin_flag,
            "flips_margin_flagged": int((flips & (margin < margin_flag)).sum()),
            "flips_outside_margin": int((flips & (margin >= margin_flag)).sum()),
            "flip_margin_max": float(margin[flips].max()) if flips.any() else 0.0,
            "queries_within_margin": int((margin < margin_flag).sum()),
            "prob_abs_err_max": float(np.abs(np.asarray(p_t, np.float64) - np.asarray(p_r, np.float64)).max()) if n else 0.0,
        })
    m = omc[same]
    q = (lambda v, x: float(np.quantile(v, x))) if len(m) else (lambda v, x: 0.0)
    out["one_minus_cos"] = {"p50": q(m, 0.5), "p99": q(m, 0.99), "max": float(m.max()) if len(m) else 0.0,
                            "over": "queries whose arg-max agrees (a flipped query is a different expert's normal)",
                            "max_incl_flips": float(omc.max()) if n else 0.0}
    out["meets_north_star"] = bool(out.get("flips_outside_margin", 0) == 0 and out["one_minus_cos"]["max"] <= cos_tol)
    return out
