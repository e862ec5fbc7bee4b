"""Deterministic synthetic point clouds with analytic normals.

PCPNet data is not available (reference .gitignore:1-2, no network), so tests,
fixtures and bench.py all draw their clouds from here.  Everything is seeded
through the legacy ``numpy.random.RandomState`` so the stream is stable across
numpy versions.
"""
import numpy as np

SHAPES = ("sphere", "ellipsoid", "torus", "box")
PCPNET_NOISE = (0.0, 0.00125, 0.006, 0.012)  # x bbdiag (BASELINE cfg 4)


def _unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def make_cloud(shape="ellipsoid", n=100000, seed=1234, noise=0.0, density=None):
    """Return (pts[n,3] float32, normals[n,3] float32).

    noise:   isotropic Gaussian sigma as a fraction of the bounding-box diagonal.
    density: None | 'gradient' | 'striped' (PCPNet varying-density sets).
    """
    rng = np.random.RandomState(seed)
    m = n if density is None else 6 * n
    if shape in ("sphere", "ellipsoid"):
        ax = np.array([1.0, 1.0, 1.0]) if shape == "sphere" else np.array([1.0, 0.7, 0.5])
        u = _unit(rng.normal(size=(m, 3)))
        pts = u * ax
        nrm = _unit(u / ax)
    elif shape == "torus":
        R, r = 1.0, 0.35
        a = rng.uniform(0, 2 * np.pi, m)
        b = rng.uniform(0, 2 * np.pi, m)
        pts = np.stack([(R + r * np.cos(b)) * np.cos(a), (R + r * np.cos(b)) * np.sin(a), r * np.sin(b)], 1)
        nrm = np.stack([np.cos(b) * np.cos(a), np.cos(b) * np.sin(a), np.sin(b)], 1)
    elif shape == "box":
        face = rng.randint(0, 6, m)
        uv = rng.uniform(-1, 1, (m, 2))
        pts = np.zeros((m, 3))
        nrm = np.zeros((m, 3))
        for f in range(6):
            sel = face == f
            axis, sign = f // 2, 1.0 if f % 2 == 0 else -1.0
            others = [a for a in range(3) if a != axis]
            pts[sel, axis] = sign
            pts[np.ix_(sel, others)] = uv[sel]
            nrm[sel, axis] = sign
    else:
        raise ValueError("unknown shape %r" % (shape,))
    if density is not None:
        t = (pts[:, 0] - pts[:, 0].min()) / (np.ptp(pts[:, 0]) + 1e-12)
        if density == "gradient":
            keep_p = 1.0 - 0.9 * t
        elif density == "striped":
            keep_p = np.where((np.floor(t * 10).astype(int) % 2) == 0, 1.0, 0.15)
        else:
            raise ValueError("unknown density %r" % (density,))
        keep = np.nonzero(rng.uniform(size=m) < keep_p)[0]
        keep = keep[rng.permutation(len(keep))[:n]]
        if len(keep) < n:
            raise RuntimeError("density resampling produced too few points")
        pts, nrm = pts[keep], nrm[keep]
    if noise > 0:
        bbdiag = np.linalg.norm(pts.max(0) - pts.min(0))
        pts = pts + rng.normal(scale=noise * bbdiag, size=pts.shape)
    return pts.astype(np.float32), nrm.astype(np.float32)
