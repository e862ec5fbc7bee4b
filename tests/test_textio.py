"""Native text I/O (csrc/textio.cpp) is byte-/bit-identical to the numpy calls of the reference."""
import time

import numpy as np


def test_write_matches_savetxt_bytes(tmp_path):
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import textio
    rng = np.random.RandomState(0)
    a = (rng.randn(500, 3) * np.array([1e-8, 1.0, 1e5])).astype(np.float32)
    a[0] = [0.0, -0.0, 1.0]
    a[1] = [np.float32(1e-38), np.float32(3.4e38), np.float32(-1.17549435e-38)]
    p1, p2 = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    np.savetxt(p1, a.astype(np.float64))
    textio.write_f32(p2, a)
    assert open(p1, "rb").read() == open(p2, "rb").read()
    probs = rng.rand(200, 7).astype(np.float32)
    np.savetxt(p1, probs.astype(np.float64))
    textio.write_f32(p2, probs)
    assert open(p1, "rb").read() == open(p2, "rb").read()
    e = rng.randint(0, 7, 300).astype(np.int32)
    np.savetxt(p1, e.astype(int), fmt="%i")
    textio.write_i32(p2, e)
    assert open(p1, "rb").read() == open(p2, "rb").read()


def test_read_matches_loadtxt_bits(tmp_path):
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import synth, textio
    pts, nrm = synth.make_cloud("torus", n=20000, seed=4, noise=0.006)
    p = str(tmp_path / "s.xyz")
    np.savetxt(p, np.concatenate([pts, nrm], 1), fmt="%.9g")         # 6 columns; the loader keeps 3
    t = time.time()
    ref = np.loadtxt(p).astype("float32")
    t_np = time.time() - t
    t = time.time()
    got = textio.read_matrix(p, take_cols=3)
    t_nat = time.time() - t
    assert got.shape == (20000, 3) and np.array_equal(got.view(np.uint32), ref[:, :3].view(np.uint32))
    full = textio.read_matrix(p)
    assert full.shape == (20000, 6) and np.array_equal(full, ref)
    print("loadtxt %.3fs native %.3fs" % (t_np, t_nat))
    # awkward but legal text: scientific notation, tabs, blank lines, comments, no trailing newline
    q = str(tmp_path / "odd.xyz")
    open(q, "w").write("# header\n1e-3\t-2.5E+2   3\n\n  4 5 6  \r\n7.000000001 8 9")
    assert np.array_equal(textio.read_matrix(q), np.loadtxt(q).astype("float32"))
    assert textio.read_matrix(q, 3).shape == (3, 3)


def test_read_errors(tmp_path):
    import nesti_net_amd  # noqa: F401
    import pytest
    from nesti_net_amd import _lib, textio
    with pytest.raises(_lib.NestiError):
        textio.read_matrix(str(tmp_path / "missing.xyz"))
    q = str(tmp_path / "ragged.xyz")
    open(q, "w").write("1 2 3\n4 5\n")
    with pytest.raises(_lib.NestiError):
        textio.read_matrix(q)
