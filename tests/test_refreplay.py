"""refreplay.cpp (host C++) against numpy itself: the pick table the reference-order patch kernel applies must be exactly what
``RandomState(seed).choice(n, P, replace=False)`` returns, ball after ball on ONE shared stream, only over-full balls drawing
(utils/pcpnet_dataset.py:237-240, 320-321).  CPU only: no GPU call."""
import ctypes

import numpy as np
import pytest

import nesti_net_amd  # noqa: F401
from nesti_net_amd import _lib
from nesti_net_amd.refsample import REFERENCE_SEED, RefStream


def _numpy_picks(seed, sizes, P):
    rng = np.random.RandomState(seed)
    return [rng.choice(int(n), P, replace=False) if n > P else None for n in sizes]


@pytest.mark.parametrize("P", [512, 7, 1])
def test_pick_table_equals_numpy_choice(P):
    rs = np.random.RandomState(5)
    sizes = np.concatenate([rs.randint(0, 4 * P + 40, size=300), [P, P + 1, 2 * P, 65535, 1, 0]]).astype(np.int32)
    want = _numpy_picks(REFERENCE_SEED, sizes, P)
    st = RefStream(REFERENCE_SEED)
    # two calls on one stream: the state carries over like the reference's shared RandomState does from batch to batch
    cut = 123
    got, off = [], []
    for part in (sizes[:cut], sizes[cut:]):
        picks, offsets = st.picks(part, P)
        got.append(picks)
        off.append(offsets)
    for part, picks, offsets, base in ((sizes[:cut], got[0], off[0], 0), (sizes[cut:], got[1], off[1], cut)):
        assert offsets.shape == part.shape
        for b, n in enumerate(part):
            w = want[base + b]
            if w is None:
                assert offsets[b] == -1
            else:
                assert np.array_equal(picks[offsets[b]:offsets[b] + P].astype(np.int64), w), (base + b, n)
    # the stream is where numpy's is: the next draw agrees too
    rng = np.random.RandomState(REFERENCE_SEED)
    for n in sizes:
        if n > P:
            rng.choice(int(n), P, replace=False)
    nxt, o = st.picks(np.array([3 * P + 5], np.int32), P)
    assert np.array_equal(nxt[o[0]:o[0] + P].astype(np.int64), rng.choice(3 * P + 5, P, replace=False))


def test_refusals_leave_the_stream_untouched():
    st = RefStream(11)
    with pytest.raises(_lib.NestiError):
        st.picks(np.array([600, 70000], np.int32), 512)          # uint16 table: more than 65535 points in a ball
    with pytest.raises(_lib.NestiError):
        st.picks(np.array([600, -1], np.int32), 512)
    picks, off = st.picks(np.array([600], np.int32), 512)
    assert np.array_equal(picks[off[0]:off[0] + 512].astype(np.int64), np.random.RandomState(11).choice(600, 512, replace=False))
    picks, off = st.picks(np.zeros(0, np.int32), 512)            # an empty batch draws nothing
    assert len(picks) == 0 and len(off) == 0


def test_traversal_order_is_the_order_of_tree_indices():
    """The fact the GPU path rests on: cKDTree.query_ball_point returns a ball in ascending position in
    ``tree.indices`` (it visits `lesser` before `greater`, and a leaf is a contiguous slice of ``tree.indices``) -- single-point
    queries (what the reference issues, utils/pcpnet_dataset.py:304) and batched ``return_sorted=False`` queries alike, on the
    fixture clouds at the fixtures' radii.  (The fixtures' ``ball_concat`` is index-sorted and says nothing about order; their
    patch ROWS do, and tests/test_gpu_patches.py holds the GPU reference-order rows to them.)"""
    from scipy import spatial
    from conftest import golden_patch_files, load_golden_patches
    for path in golden_patch_files()[:3]:
        g = load_golden_patches(path)
        tree = spatial.cKDTree(g["pts"], 10)                      # utils/pcpnet_dataset.py:37
        rank = np.empty(len(g["pts"]), np.int64)
        rank[tree.indices] = np.arange(len(g["pts"]))
        c = g["pts"][:: max(1, len(g["pts"]) // 50)]
        for rad in g["r_abs"]:
            for i, ball in enumerate(tree.query_ball_point(c, float(rad), return_sorted=False)):
                assert np.all(np.diff(rank[np.asarray(ball, np.int64)]) > 0)
                if i < 8:
                    one = tree.query_ball_point(c[i], float(rad))
                    assert one == ball


def test_prefetch_stops_its_worker_when_the_consumer_fails():
    """pipeline.NormalEstimator._prefetch (the producer / consumer of both reference-order paths): whatever fails, the
    worker thread is stopped and joined before the error propagates, and the estimator refuses further reference-order runs --
    the shared random stream would no longer line up with the reference's."""
    import threading
    import types
    from nesti_net_amd.pipeline import NormalEstimator
    made, consumed = [], []

    def make(i, span):
        made.append(i)
        return i * 10

    def consume_ok(i, span, item, release):
        consumed.append((i, item))
        release()

    me = types.SimpleNamespace(_ref_failed=False)
    before = threading.active_count()
    NormalEstimator._prefetch(me, list(range(5)), make, consume_ok)
    assert consumed == [(i, i * 10) for i in range(5)] and not me._ref_failed and threading.active_count() == before

    def consume_bad(i, span, item, release):
        release()
        if i == 1:
            raise RuntimeError("upload failed")

    made.clear()
    with pytest.raises(RuntimeError, match="upload failed"):
        NormalEstimator._prefetch(me, list(range(50)), make, consume_bad)
    assert me._ref_failed and threading.active_count() == before and len(made) < 50      # the producer did not run on

    def make_bad(i, span):
        if i == 2:
            raise ValueError("sampler failed")
        return i

    me2 = types.SimpleNamespace(_ref_failed=False)
    with pytest.raises(ValueError, match="sampler failed"):
        NormalEstimator._prefetch(me2, list(range(6)), make_bad, consume_ok)
    assert me2._ref_failed and threading.active_count() == before
