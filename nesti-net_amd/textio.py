"""Fast text I/O of the file seam (native code in ``csrc/textio.cpp``), byte-compatible with the
numpy calls the reference uses: ``np.loadtxt(...).astype('float32')`` for ``<shape>.xyz``
(``utils/pcpnet_dataset.py:250``) and ``np.savetxt`` for ``.normals`` / ``.experts`` /
``.experts_probs`` (``test_n_est_w_experts.py:182-188``)."""
import ctypes

import numpy as np

from . import _lib


def read_matrix(path, take_cols=None):
    """Whitespace-separated numeric text -> float32 [rows, take_cols or all columns]."""
    lib = _lib.load()
    n, c = ctypes.c_int64(0), ctypes.c_int(0)
    bpath = path.encode()
    _lib.check(lib.nesti_read_text_matrix(bpath, None, 0, 0, ctypes.byref(n), ctypes.byref(c)), "nesti_read_text_matrix")
    cols = c.value if take_cols is None else int(take_cols)
    out = np.empty((n.value, cols), np.float32)
    if n.value:
        _lib.check(lib.nesti_read_text_matrix(bpath, _lib.ptr(out), n.value, cols, ctypes.byref(n), ctypes.byref(c)),
                   "nesti_read_text_matrix")
    return out


def write_f32(path, a):
    """Same bytes as ``np.savetxt(path, a.astype(np.float64))``."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    a2 = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(-1, 1)
    _lib.check(_lib.load().nesti_write_text_f32(path.encode(), _lib.ptr(a2), a2.shape[0], a2.shape[1]), "nesti_write_text_f32")


def write_i32(path, a):
    """Same bytes as ``np.savetxt(path, a.astype(int), fmt='%i')``."""
    a = np.ascontiguousarray(a, dtype=np.int32).ravel()
    _lib.check(_lib.load().nesti_write_text_i32(path.encode(), _lib.ptr(a), a.shape[0]), "nesti_write_text_i32")
