"""Pins oracle/patches_ref.py against patches captured from the reference's own
PointcloudPatchDataset (tests/golden/patches_*.npz, scripts/make_golden_patches.py)."""
import numpy as np
import pytest

from conftest import golden_patch_files, load_golden_patches
from oracle import patches_ref


def _rows_sorted(a):
    a = np.ascontiguousarray(a)
    v = a.view([("x", a.dtype), ("y", a.dtype), ("z", a.dtype)]).ravel()
    return np.sort(v, order=["x", "y", "z"])


@pytest.mark.parametrize("path", golden_patch_files(), ids=lambda p: p.split("patches_")[-1][:-4])
def test_oracle_matches_reference_patches(path):
    g = load_golden_patches(path)
    pts, P = g["pts"], g["P"]
    bbdiag, r_abs = patches_ref.patch_radii(pts, list(g["radii"]))
    assert np.array_equal(np.asarray(r_abs), g["r_abs"])          # utils/pcpnet_dataset.py:281-282
    points, n_eff, nbr, n_ball = patches_ref.extract_patches(pts, g["queries"], r_abs, P, g["seed"])
    assert np.array_equal(n_eff, g["n_eff"])                       # exact: integer
    M, S = n_eff.shape
    capped = 0
    for q in range(M):
        for s in range(S):
            ball = g["balls"][q][s]
            assert n_ball[q, s] == len(ball)
            mine = nbr[q, s * P:s * P + n_eff[q, s]]
            assert np.all(nbr[q, s * P + n_eff[q, s]:(s + 1) * P] == -1)
            ref_rows = g["points"][q, s * P:(s + 1) * P]
            my_rows = points[q, s * P:(s + 1) * P]
            assert np.all(my_rows[n_eff[q, s]:] == 0) and np.all(ref_rows[n_eff[q, s]:] == 0)
            if len(ball) <= P:
                assert np.array_equal(np.sort(mine), ball)
                # same multiset of rows, bit for bit (order differs: cKDTree traversal vs key order)
                assert np.array_equal(_rows_sorted(my_rows[:n_eff[q, s]]), _rows_sorted(ref_rows[:n_eff[q, s]]))
            else:
                capped += 1
                assert len(np.unique(mine)) == P and np.all(np.isin(mine, ball))
                # the reference's rows are P members of the same ball under the same f32 arithmetic
                center = pts[g["queries"][q]]
                all_rows = (pts[ball] - center) / np.float32(r_abs[s])
                assert np.all(np.isin(_rows_sorted(ref_rows), _rows_sorted(all_rows)))
                assert np.all(np.isin(_rows_sorted(my_rows), _rows_sorted(all_rows)))
    if "100k" in path or "smallP" in path:
        assert capped > 0          # the random-subsample branch (:320-321) is exercised


def test_subsample_is_uniform_and_deterministic():
    idx = np.arange(5000)
    h1 = patches_ref.subsample_hash(3627473, 7, 2, idx)
    h2 = patches_ref.subsample_hash(3627473, 7, 2, idx)
    assert np.array_equal(h1, h2)
    assert not np.array_equal(h1, patches_ref.subsample_hash(3627473, 8, 2, idx))
    # roughly uniform over 32 bits
    hist, _ = np.histogram(h1, bins=8, range=(0, 2 ** 32))
    assert hist.min() > 500
    # known answers (pins the hash against csrc/patches.hip)
    assert [int(x) for x in patches_ref.subsample_hash(3627473, 0, 0, np.arange(3))] == [3653716589, 2655192765, 1943666080]
    assert [int(x) for x in patches_ref.subsample_hash(3627473, 99999, 2, np.array([0, 77777]))] == [3736329315, 64764641]
