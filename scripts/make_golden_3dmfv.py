"""Generate tests/golden/fv_numpy_ref.npz by running the REFERENCE's own numpy 3DmFV and grid-GMM code
(/root/reference/utils/utils.py: get_3DmFV :260-332, get_3d_grid_gmm :70-95) on patches from the committed golden
patch fixtures.  Runs only in the build container (the reference tree does not travel to the GPU box).

Two import shims, neither of which touches the code under test:
  * ``h5py`` (absent here) is imported by utils/provider.py, which utils/utils.py imports at module level but does
    not use in these two functions -> an empty placeholder module;
  * utils.get_3d_grid_gmm imports ``sklearn.mixture.gaussian_mixture._compute_precision_cholesky`` (the pre-0.22
    module path) -> aliased to the REAL function of the installed scikit-learn (``sklearn.mixture._gaussian_mixture``).

    python scripts/make_golden_3dmfv.py
"""
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, "/root/reference/utils")

sys.modules.setdefault("h5py", types.ModuleType("h5py"))
import sklearn.mixture._gaussian_mixture as _gm  # noqa: E402
sys.modules.setdefault("sklearn.mixture.gaussian_mixture", _gm)

import utils as ref_utils  # noqa: E402  (the reference)
from conftest import golden_patch_files, load_golden_patches  # noqa: E402

out = {}
for n, var in ((8, 0.0156), (3, 0.111)):
    gmm = ref_utils.get_3d_grid_gmm(subdivisions=[n, n, n], variance=var)
    out["gmm%d_weights" % n], out["gmm%d_means" % n], out["gmm%d_covariances" % n] = gmm.weights_, gmm.means_, gmm.covariances_
    g = load_golden_patches([p for p in golden_patch_files() if "ellipsoid20k" in p][0])
    P = int(g["P"])
    rows = [(1, 0), (3, 1), (5, 2), (20, 1)]                       # (query row, scale) of the fixture
    pts = np.stack([g["points"][q, s * P:(s + 1) * P] for q, s in rows]).astype(np.float64)
    fv = ref_utils.get_3DmFV(pts, gmm.weights_, gmm.means_, np.sqrt(gmm.covariances_), normalize=True)
    assert fv.shape == (len(rows), 20, n ** 3)
    out["fv%d" % n] = fv
    out["fv%d_rows" % n] = np.asarray(rows, np.int32)
    fv_raw = ref_utils.get_3DmFV(pts[:2], gmm.weights_, gmm.means_, np.sqrt(gmm.covariances_), normalize=False)
    out["fv%d_raw" % n] = fv_raw
path = os.path.join(REPO, "tests", "golden", "fv_numpy_ref.npz")
np.savez_compressed(path, **out)
print(path, "%.1f KB" % (os.path.getsize(path) / 1024), {k: v.shape for k, v in out.items()})
