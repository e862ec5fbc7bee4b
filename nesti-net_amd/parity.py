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

ONE tie rule everywhere (VERDICT / ADVICE r04): the test suite (tests/test_gpu_fixtures.py, tests/test_gpu_net.py),
``__graft_entry__.smoke()`` and the bench all decide pass / fail with ``TIE_MARGIN``.  The hand-picked 2e-5 of rounds 1-3
(``TIE_MARGIN_HAND``) is kept as a REPORTED band only: ``flips_gap_hand_to_margin`` counts the differences whose reference
gap lies in [2e-5, TIE_MARGIN) -- excused by the derived rule, not by the old one -- and ``meets_north_star_at_2e-5`` is the
verdict the old constant would give.  :func:`adjudicate` puts the question to the fp64 oracle's own probabilities for exactly
the differing queries (the caller computes them; this module never imports ``oracle/``).
"""
import numpy as np

COS_TOL = 1e-5            # north star: cosine tolerance on the normal vectors
# |p_f32mode - p_fp64oracle| over all queries of the six fixture clouds (tests/test_gpu_fixtures.py asserts <= the bound on
# every fixture and prints the measured value; the arithmetic is deterministic, so the figure is the same on every box)
F32_PROB_ERR_MEASURED = 6.0e-5
F32_PROB_ERR_BOUND = 6.5e-5
TIE_MARGIN = 2 * F32_PROB_ERR_BOUND   # the fp32 reference's own top-2 probabilities closer than this: fp32 arithmetic does not define the arg-max
TIE_MARGIN_HAND = 2e-5                # rounds 1-3's hand-picked margin: reported next to the derived one, never the pass / fail rule


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
            "tie_margin_hand": TIE_MARGIN_HAND,
            "flips_gap_hand_to_margin": int((flips & (margin >= TIE_MARGIN_HAND) & (margin < tie_margin)).sum()),
            "flip_rows": np.nonzero(flips)[0][:64].tolist(),
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
    out["meets_north_star_at_2e-5"] = bool(out["meets_north_star"] and out.get("flips_gap_hand_to_margin", 0) == 0)
    out["strict_bit_exact_argmax"] = bool(out.get("argmax_flips", 0) == 0)
    return out


def adjudicate(rows, e_test, e_ref, p_ref, p_oracle, tie_margin=TIE_MARGIN):
    """The fp64 oracle's verdict on the queries where ``test`` and the f32-mode reference disagree.

    ``rows``: their row numbers; ``e_test`` / ``e_ref`` [k]: the two arg-maxes; ``p_ref`` [k,E]: the f32 mode's
    probabilities; ``p_oracle`` [k,E]: the fp64 oracle's probabilities of the same queries.  Per query: the oracle's own
    top-2 gap, its arg-max, and whose side it takes.  A difference is *adjudicated as a tie* when the oracle's gap is below
    ``tie_margin`` too: the exact network itself then sits within fp32 evaluation noise of a tie, whichever side the oracle
    happens to land on.  ``all_ties`` is what ``meets_north_star`` additionally requires when the adjudication ran."""
    p_o = np.asarray(p_oracle, np.float64).reshape(len(rows), -1)
    p_r = np.asarray(p_ref, np.float64).reshape(len(rows), -1)
    per = []
    for i, row in enumerate(rows):
        so, sr = np.sort(p_o[i]), np.sort(p_r[i])
        e_o = int(np.argmax(p_o[i]))
        per.append({"row": int(row), "expert_test": int(e_test[i]), "expert_f32_mode": int(e_ref[i]), "expert_oracle_fp64": e_o,
                    "gap_f32_mode": float(sr[-1] - sr[-2]), "gap_oracle_fp64": float(so[-1] - so[-2]),
                    "oracle_sides_with": "test" if e_o == int(e_test[i]) else "f32_mode" if e_o == int(e_ref[i]) else "neither"})
    gaps = [d["gap_oracle_fp64"] for d in per]
    return {"flips": per, "flips_oracle_sides_with_test": sum(d["oracle_sides_with"] == "test" for d in per),
            "flips_oracle_sides_with_f32_mode": sum(d["oracle_sides_with"] == "f32_mode" for d in per),
            "oracle_gap_max": max(gaps) if gaps else 0.0, "tie_margin": tie_margin,
            "all_ties": bool(all(g < tie_margin for g in gaps)),
            "all_ties_at_2e-5": bool(all(g < TIE_MARGIN_HAND for g in gaps)),
            "note": "fp64 CPU oracle (oracle/net_ref.gate_forward on oracle patches + MuPS) evaluated on exactly the queries "
                    "whose arg-max differs from the f32 mode's, outside the timed region"}
