"""Fast text output of the file seam (native code in ``csrc/textio.cpp``), byte-identical to the ``np.savetxt`` calls
the reference uses for ``.normals`` / ``.experts`` / ``.experts_probs`` (``test_n_est_w_experts.py:182-188``).  Reading
``<shape>.xyz`` stays ``np.loadtxt`` + the ``.npy`` cache (``provider.load_xyz``, ``utils/pcpnet_dataset.py:249-251``)."""
import numpy as np

from . import _lib


def write_f32(path, a):
    """Same bytes as ``np.savetxt(path, a.astype(np.float64))``."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    a2 = a.reshape(a.shape[0], -1) if a.ndim > 1 else a.reshape(-1, 1)
    _lib.check(_lib.load().nesti_write_text_f32(path.encode(), _lib.ptr(a2), a2.shape[0], a2.shape[1]), "nesti_write_text_f32")


def write_i32(path, a):
    """Same bytes as ``np.savetxt(path, a.astype(int), fmt='%i')``."""
    a = np.ascontiguousarray(a, dtype=np.int32).ravel()
    _lib.check(_lib.load().nesti_write_text_i32(path.encode(), _lib.ptr(a), a.shape[0]), "nesti_write_text_i32")
