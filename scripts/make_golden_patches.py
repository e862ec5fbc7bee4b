"""Generate tests/golden/patches_*.npz by running the REFERENCE's own patch dataset
(/root/reference/utils/pcpnet_dataset.py, importable under py3) on synthetic clouds.

Runs only in the build container (the reference tree does not travel to the GPU box);
the committed .npz files are plain data: inputs (cloud recipe, radii, query indices) and
the reference's outputs (patch tensors, n_eff) plus the uncapped ball sets taken from the
reference dataset's own cKDTree.

    python scripts/make_golden_patches.py
"""
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/utils")

import pcpnet_dataset  # noqa: E402  (the reference)
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import synth  # noqa: E402

SEED = 3627473  # test_n_est_w_experts.py:113
CASES = [
    # name, cloud kwargs, radii, P, query indices
    dict(name="sphere8k_ss", cloud=dict(shape="sphere", n=8000, seed=1234), radii=[0.05], P=512,
         queries=list(range(16)) + list(range(100, 8000, 500))),
    dict(name="ellipsoid20k_noise", cloud=dict(shape="ellipsoid", n=20000, seed=1235, noise=0.006),
         radii=[0.01, 0.03, 0.05], P=512, queries=list(range(12)) + list(range(50, 20000, 1000))),
    dict(name="ellipsoid100k", cloud=dict(shape="ellipsoid", n=100000, seed=1234), radii=[0.01, 0.03, 0.05], P=512,
         queries=list(range(16)) + list(range(777, 100000, 6250))),
    dict(name="box20k_smallP", cloud=dict(shape="box", n=20000, seed=1236), radii=[0.02, 0.06], P=64,
         queries=list(range(8)) + list(range(31, 20000, 2500))),
    # PCPNet's varying-density sets and strongest noise level (BASELINE configs 3 / 4)
    dict(name="torus40k_gradient", cloud=dict(shape="torus", n=40000, seed=1237, noise=0.00125, density="gradient"),
         radii=[0.01, 0.03, 0.05], P=512, queries=list(range(6)) + list(range(123, 40000, 4000))),
    dict(name="sphere40k_striped_noise012", cloud=dict(shape="sphere", n=40000, seed=1238, noise=0.012, density="striped"),
         radii=[0.01, 0.03, 0.05], P=512, queries=list(range(6)) + list(range(77, 40000, 4000))),
]


def run_case(case, outdir):
    pts, _ = synth.make_cloud(**case["cloud"])
    with tempfile.TemporaryDirectory() as d:
        np.savetxt(os.path.join(d, "shape0.xyz"), pts, fmt="%.9g")
        with open(os.path.join(d, "list.txt"), "w") as f:
            f.write("shape0\n")
        np.savetxt(os.path.join(d, "shape0.pidx"), np.asarray(case["queries"]), fmt="%d")
        ds = pcpnet_dataset.PointcloudPatchDataset(
            root=d, shape_list_filename="list.txt", patch_radius=case["radii"], points_per_patch=case["P"],
            patch_features=[], seed=SEED, identical_epochs=False, use_pca=False, center="point", point_tuple=1,
            cache_capacity=100, point_count_std=0, sparse_patches=True)   # utils/provider.py:389-425 wiring
        shape = ds.shape_cache.get(0)
        assert np.array_equal(shape.pts, pts), "text round trip changed the cloud"
        M = len(case["queries"])
        S, P = len(case["radii"]), case["P"]
        points = np.zeros((M, S * P, 3), np.float32)
        n_eff = np.zeros((M, S), np.int32)
        for i in range(M):                      # SequentialPointcloudPatchSampler order
            item = ds[i]
            points[i] = item[0].numpy()
            n_eff[i] = np.asarray(item[-1]).astype(np.int32)
        r_abs = np.asarray(ds.patch_radius_absolute[0], np.float64)
        sets, offs = [], [0]
        for c in case["queries"]:
            for rad in r_abs:
                b = np.sort(np.asarray(shape.kdtree.query_ball_point(shape.pts[c, :], float(rad)), np.int32))
                sets.append(b)
                offs.append(offs[-1] + len(b))
    path = os.path.join(outdir, "patches_%s.npz" % case["name"])
    np.savez_compressed(
        path, cloud_shape=case["cloud"]["shape"], cloud_n=case["cloud"]["n"], cloud_seed=case["cloud"]["seed"],
        cloud_noise=case["cloud"].get("noise", 0.0), cloud_density=case["cloud"].get("density") or "", radii=np.asarray(case["radii"]), P=P, seed=SEED,
        queries=np.asarray(case["queries"], np.int32), r_abs=r_abs, points=points, n_eff=n_eff,
        ball_concat=np.concatenate(sets).astype(np.int32), ball_offsets=np.asarray(offs, np.int64))
    print(path, "M=%d" % M, "n_eff mean", n_eff.mean(0), "ball mean",
          np.diff(offs).reshape(M, S).mean(0), "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    out = os.path.join(REPO, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    only = set(sys.argv[1:])          # optional: names of the cases to (re)generate
    for c in CASES:
        if not only or c["name"] in only:
            run_case(c, out)
