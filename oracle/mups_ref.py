"""ORACLE (test infrastructure, NOT product code): CPU restatement of the
reference's MuPS / 3DmFV arithmetic.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``nesti-net_amd/``) never does.

Parity status: **partly pinned**.  TensorFlow 1.12 is not installable here, so
``utils/tf_util.py:655-753`` (get_3dmfv_n_est) itself cannot be executed.  ``mups_literal`` is a
line-by-line numpy transcription of it.  The reference also ships a numpy twin of the same
construction, ``utils/utils.py:260-330`` (get_3DmFV): identical Gaussian density, derivative
terms, scale constants, max/min/sum reductions, power + L2 normalisation and channel order; it
differs from the TF function in exactly four places (no ``w p / sum(w p)`` posterior, no padding
mask, division by the row count instead of ``n_eff``, L2 norm without epsilon).  With those four
switched through keyword flags, ``mups_literal`` reproduces the outputs of the reference's numpy
function run in this container (``scripts/make_golden_3dmfv.py`` -> ``tests/golden/fv_numpy_ref.npz``)
to 1e-13, and ``grid_gmm`` equals its ``get_3d_grid_gmm`` exactly (``tests/test_oracle_mups.py``).
The four TF-only lines (``:693-702``, ``:722-729``, ``:738-740``) remain transcription-only.
``mups_separable`` is an independent derivation (SURVEY.md §8(a')) cross-checked against the literal form.

All file:line citations are relative to the reference tree.
"""
import os

import numpy as np


def grid_gmm(n=8, variance=0.0156):
    """``utils/utils.py:70-95`` get_3d_grid_gmm, minus the sklearn wrapper.

    Returns (w[G], mu[G,3], sigma[G,3]) as float64; sigma = sqrt(covariances_)
    as fed at ``test_n_est_w_experts.py:146``.  x is the slowest axis."""
    step = 1.0 / n
    means = np.mgrid[step - 1:1.0 - step:complex(0, n),
                     step - 1:1.0 - step:complex(0, n),
                     step - 1:1.0 - step:complex(0, n)]
    means = np.reshape(means, [3, -1]).T
    cov = variance * np.ones_like(means)
    w = (1.0 / n ** 3) * np.ones(n ** 3)
    return w, means, np.sqrt(cov)


def mups_literal(points, w, mu, sigma, n_eff, dtype=np.float64, posterior=True, masked=True, per_n_eff=True,
                 l2_eps=1e-12, flatten=True):
    """One scale.  Transcribes ``utils/tf_util.py:655-753`` (flatten=True).

    points [B,P,3], n_eff [B] -> fv [B, 20*G].

    The keyword flags exist only to pin this transcription against the reference's numpy twin
    ``utils/utils.py:260-330``: ``posterior=False`` (Q = p, ``:300-301`` there), ``masked=False`` (no padding
    mask), ``per_n_eff=False`` (divide by the row count, ``:313-315``), ``l2_eps=0`` (``l2_normalize``,
    ``:247-255``) and ``flatten=False`` ([B,20,G], ``:331-332``) make it that function."""
    points = np.asarray(points, dtype)
    w = np.asarray(w, dtype)
    mu = np.asarray(mu, dtype)
    sigma = np.asarray(sigma, dtype)
    n_orig = np.asarray(n_eff).astype(np.int32)            # :664
    B, P, D = points.shape
    G = mu.shape[0]
    batch_sig = np.broadcast_to(sigma[None, None], (B, P, G, D))      # :671-672
    batch_mu = np.broadcast_to(mu[None, None], (B, P, G, D))          # :673-674
    batch_w = np.broadcast_to(w[None, None], (B, P, G))               # :675
    batch_points = np.broadcast_to(points[:, :, None, :], (B, P, G, D))  # :676
    w_per_batch_per_d = np.broadcast_to(w[None, :, None], (B, G, 3 * D))  # :680
    z = (batch_points - batch_mu) / batch_sig
    p_per_point = (dtype(1.0) / (np.power(dtype(2.0 * np.pi), dtype(D / 2.0)) *
                                 np.power(batch_sig[..., 0], dtype(D)))) * \
        np.exp(dtype(-0.5) * np.sum(np.square(z), axis=3))             # :686-687
    r = np.arange(P)[None, :, None]                                    # :690-692
    mask = r > n_orig[:, None, None]                                   # :693-695  (NOT >=)
    mask = np.broadcast_to(mask, (B, P, G)) if masked else np.zeros((B, P, G), bool)
    w_zero_comp = np.where(mask, batch_w, dtype(0))                    # :697
    w_p = p_per_point * batch_w                                        # :699
    Q = w_p / np.sum(w_p, axis=-1, keepdims=True) if posterior else p_per_point   # :700
    Q = np.where(mask, dtype(0), Q)                                    # :702
    Qd = Q[..., None]                                                  # :703
    d_pi_all = ((Q - batch_w + w_zero_comp) / np.sqrt(batch_w))[..., None]   # :709
    d_pi = np.concatenate([d_pi_all.max(1), d_pi_all.sum(1)], axis=2)  # :710-711
    d_mu_all = Qd * z                                                  # :713
    d_mu = (1 / np.sqrt(w_per_batch_per_d)) * np.concatenate(
        [d_mu_all.max(1), d_mu_all.min(1), d_mu_all.sum(1)], axis=2)   # :714-715
    d_sig_all = Qd * (np.power(z, 2) - 1)                              # :717
    d_sigma = (1 / np.sqrt(2 * w_per_batch_per_d)) * np.concatenate(
        [d_sig_all.max(1), d_sig_all.min(1), d_sig_all.sum(1)], axis=2)  # :718-719
    with np.errstate(divide="ignore", invalid="ignore"):
        eff = n_orig.astype(np.float32).astype(dtype)[:, None, None] if per_n_eff else dtype(P)   # :722
        d_pi, d_mu, d_sigma = d_pi / eff, d_mu / eff, d_sigma / eff    # :727-729
    alpha = dtype(0.5)                                                 # :732-735
    d_pi = np.sign(d_pi) * np.power(np.abs(d_pi), alpha)
    d_mu = np.sign(d_mu) * np.power(np.abs(d_mu), alpha)
    d_sigma = np.sign(d_sigma) * np.power(np.abs(d_sigma), alpha)

    def l2n(x):  # tf.nn.l2_normalize(axis=1): x * rsqrt(max(sum(x^2), 1e-12))   :738-740
        return x / np.sqrt(np.maximum(np.sum(np.square(x), axis=1, keepdims=True), dtype(l2_eps)))
    d_pi, d_mu, d_sigma = l2n(d_pi), l2n(d_mu), l2n(d_sigma)
    if not flatten:
        return np.transpose(np.concatenate([d_pi, d_mu, d_sigma], axis=2), (0, 2, 1))   # :749-750
    flat = lambda x: np.transpose(x, (0, 2, 1)).reshape(B, -1)         # :744-746
    return np.concatenate([flat(d_pi), flat(d_mu), flat(d_sigma)], axis=1)  # :747


def mups_assemble(points, n_eff, n_scales, grid_n=8, variance=0.0156, dtype=np.float64,
                  fn=None, chunk=4, workers=None):
    """``models/experts_n_est.py:59-76``: per-scale 3DmFV -> reshape [B,20,R,R,R]
    -> transpose [B,R,R,R,20] -> concat on channels.  Returns [B,R,R,R,20*S]."""
    fn = fn or mups_literal
    w, mu, sig = grid_gmm(grid_n, variance)
    points = np.asarray(points)
    n_eff = np.asarray(n_eff)
    B = points.shape[0]
    P = points.shape[1] // n_scales
    out = np.empty((B, grid_n, grid_n, grid_n, 20 * n_scales), dtype)

    def one(b0):
        sl = slice(b0, min(B, b0 + chunk))
        for s in range(n_scales):
            fv = fn(points[sl, s * P:(s + 1) * P], w, mu, sig, n_eff[sl, s], dtype=dtype)
            fv = fv.reshape(fv.shape[0], -1, grid_n, grid_n, grid_n)   # :71
            out[sl, ..., 20 * s:20 * (s + 1)] = np.transpose(fv, (0, 2, 3, 4, 1))  # :72
    # the chunks are independent (every query's rows are its own): a few threads over them change nothing but the wall time
    # (numpy releases the GIL inside the large elementwise passes; ~0.3 GB of temporaries per thread)
    workers = max(1, min(workers if workers is not None else 16, os.cpu_count() or 1, (B + chunk - 1) // chunk))
    if workers == 1:
        for b0 in range(0, B, chunk):
            one(b0)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(one, range(0, B, chunk)))
    return out


def mups_separable(points, w, mu, sigma, n_eff, dtype=np.float64):
    """Independent derivation (SURVEY.md §8(a')): on the uniform product grid the
    posterior factorises per axis, Q_ijk = qx_i qy_j qz_k.  Same signature and
    output layout as :func:`mups_literal`; used to cross-check it."""
    points = np.asarray(points, dtype)
    B, P, _ = points.shape
    G = mu.shape[0]
    n = int(round(G ** (1.0 / 3.0)))
    a = np.asarray(mu, dtype)[::n * n, 0]          # axis grid (x slowest)
    sg = dtype(np.asarray(sigma).flat[0])
    wv = dtype(np.asarray(w).flat[0])
    m = np.asarray(n_eff).astype(np.int64)
    valid = (np.arange(P)[None, :] <= m[:, None])                       # row m IS counted
    d = (points[:, :, :, None] - a[None, None, None, :]) / sg          # [B,P,3,n]
    e = np.exp(dtype(-0.5) * d * d)
    q = e / e.sum(-1, keepdims=True)
    Q = (q[:, :, 0, :, None, None] * q[:, :, 1, None, :, None] * q[:, :, 2, None, None, :])
    Q = Q * valid[:, :, None, None, None]
    shp = (B, P, n, n, n)
    dd = [np.broadcast_to(d[:, :, 0, :, None, None], shp),
          np.broadcast_to(d[:, :, 1, None, :, None], shp),
          np.broadcast_to(d[:, :, 2, None, None, :], shp)]
    ch = []
    pi = (Q - wv * valid[:, :, None, None, None]) / np.sqrt(wv)
    ch += [pi.max(1), pi.sum(1)]
    mu_all = [Q * x for x in dd]
    ch += [x.max(1) / np.sqrt(wv) for x in mu_all]
    ch += [x.min(1) / np.sqrt(wv) for x in mu_all]
    ch += [x.sum(1) / np.sqrt(wv) for x in mu_all]
    sg_all = [Q * (x * x - 1) for x in dd]
    ch += [x.max(1) / np.sqrt(2 * wv) for x in sg_all]
    ch += [x.min(1) / np.sqrt(2 * wv) for x in sg_all]
    ch += [x.sum(1) / np.sqrt(2 * wv) for x in sg_all]
    v = np.stack(ch, axis=1).reshape(B, 20, G)
    with np.errstate(divide="ignore", invalid="ignore"):
        v = v / m.astype(dtype)[:, None, None]
    v = np.sign(v) * np.sqrt(np.abs(v))
    v = v / np.sqrt(np.maximum((v * v).sum(-1, keepdims=True), dtype(1e-12)))
    return v.reshape(B, 20 * G)
