"""The opt-in reference-order subsample (nesti_net_amd/refsample.py) against the reference's own patch tensors: the golden
fixtures were captured from /root/reference's PointcloudPatchDataset visiting their queries in order with seed 3627473
(scripts/make_golden_patches.py), balls larger than P included -- every row must come out bit for bit."""
import numpy as np
import pytest

from conftest import golden_patch_files, load_golden_patches


@pytest.mark.parametrize("native", [False, True], ids=["numpy_stream", "native_stream"])
@pytest.mark.parametrize("path", golden_patch_files(), ids=lambda p: p.split("patches_")[-1][:-4])
def test_reference_subsample_reproduces_the_golden_rows(path, native):
    """``native``: the picks come from csrc/refreplay.cpp (RefStream) instead of numpy's RandomState -- the stream the GPU
    reference-order path shares with this host path."""
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd.refsample import ReferencePatchSampler, RefStream
    g = load_golden_patches(path)
    smp = ReferencePatchSampler(seed=g["seed"], stream=RefStream(g["seed"]) if native else None)
    tree = smp.build_tree(g["pts"])
    points, n_eff = smp.patches(g["pts"], tree, g["queries"].astype(np.int64), [float(r) for r in g["r_abs"]], g["P"])
    assert np.array_equal(n_eff, g["n_eff"])
    capped = sum(len(b) > g["P"] for balls in g["balls"] for b in balls)
    assert np.array_equal(points.view(np.uint32), g["points"].view(np.uint32)), "%d capped balls" % capped
    if "100k" in path or "gradient" in path:
        assert capped > 0          # the fixtures that exercise rng.choice
