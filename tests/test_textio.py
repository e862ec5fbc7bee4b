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
