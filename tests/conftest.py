import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_patch_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "patches_*.npz")))


def load_golden_patches(path):
    """Golden patch fixture -> dict, with the synthetic cloud regenerated from its recipe."""
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import synth
    g = dict(np.load(path))
    pts, nrm = synth.make_cloud(shape=str(g["cloud_shape"]), n=int(g["cloud_n"]), seed=int(g["cloud_seed"]),
                                noise=float(g["cloud_noise"]),
                                density=(str(g["cloud_density"]) or None) if "cloud_density" in g else None)
    g["pts"], g["normals"] = pts, nrm
    g["P"], g["seed"] = int(g["P"]), int(g["seed"])
    M, S = g["n_eff"].shape
    offs = g["ball_offsets"]
    g["balls"] = [[g["ball_concat"][offs[q * S + s]:offs[q * S + s + 1]] for s in range(S)] for q in range(M)]
    return g


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
