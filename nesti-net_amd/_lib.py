"""ctypes binding of ``libnesti_hip.so`` (C-ABI: include/nesti_hip.h).

Fails loudly when the library is missing -- there is no CPU or eager fallback."""
import ctypes
import os

from .config import CConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NESTI_LIB") or os.path.join(_HERE, "libnesti_hip.so")   # NESTI_LIB: another build of the same library (A/B runs)


class CTensor(ctypes.Structure):
    """Mirror of ``nesti_tensor_t``."""
    _fields_ = [("name", ctypes.c_char_p),
                ("data", ctypes.c_void_p),
                ("ndim", ctypes.c_int),
                ("dims", ctypes.c_int64 * 5)]


class CShapeQueries(ctypes.Structure):
    """Mirror of ``nesti_shape_queries_t``."""
    _fields_ = [("cloud_dev", ctypes.c_void_p), ("n_points", ctypes.c_int),
                ("query_idx_dev", ctypes.c_void_p), ("n_queries", ctypes.c_int),
                ("r_abs", ctypes.c_double * 4), ("seed", ctypes.c_uint64), ("query_row0", ctypes.c_int),
                ("grid_ws_dev", ctypes.c_void_p), ("grid_ws_bytes", ctypes.c_size_t)]


class CCascadeStats(ctypes.Structure):
    """Mirror of ``nesti_cascade_stats_t``."""
    _fields_ = [("queries", ctypes.c_uint64), ("rechecked", ctypes.c_uint64), ("changed", ctypes.c_uint64),
                ("max_margin_err", ctypes.c_float), ("tau", ctypes.c_float),
                ("sum_sq_pair_err", ctypes.c_double), ("pairs", ctypes.c_uint64),
                ("widened", ctypes.c_uint64), ("widen_events", ctypes.c_uint64), ("tau_eff", ctypes.c_float)]


class CX8GuardStats(ctypes.Structure):
    """Mirror of ``nesti_x8_guard_stats_t``."""
    _fields_ = [("queries", ctypes.c_uint64), ("rechecked", ctypes.c_uint64), ("dropped", ctypes.c_uint64),
                ("max_dn", ctypes.c_float), ("thr", ctypes.c_float), ("thr_eff", ctypes.c_float)]


X8_GUARD_BAR, X8_GUARD_WIDEN, X8_GUARD_DEFAULT = 2.5e-6, 1.5, 0.25   # NESTI_X8_GUARD_* (include/nesti_hip.h)
GATE_WIDEN = 1.5      # NESTI_GATE_WIDEN (include/nesti_hip.h)
GATE_WIDEN_PASSES = 3  # NESTI_GATE_WIDEN_PASSES


class NestiError(RuntimeError):
    pass


_vp, _i, _sz, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_uint64
_cfgp = ctypes.POINTER(CConfig)

# name -> (restype, argtypes); every symbol include/nesti_hip.h declares
SIGNATURES = {
    "nesti_last_error": (ctypes.c_char_p, []),
    "nesti_version": (ctypes.c_char_p, []),
    "nesti_default_config": (None, [_cfgp]),
    "nesti_gmm_grid": (_i, [_i, ctypes.c_double, _vp, _vp, _vp]),
    "nesti_mups_forward": (_i, [_cfgp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "nesti_patches_workspace_bytes": (_sz, [_i]),
    "nesti_patches_grid": (_i, [_cfgp, _vp, _i, ctypes.POINTER(ctypes.c_double), _vp, _sz, _vp]),
    "nesti_patches_query": (_i, [_cfgp, _vp, _i, _vp, _i, ctypes.POINTER(ctypes.c_double), _u64, _i,
                                 _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "nesti_patches_build": (_i, [_cfgp, _vp, _i, _vp, _i, ctypes.POINTER(ctypes.c_double), _u64, _i,
                                 _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "nesti_patches_count": (_i, [_cfgp, _vp, _i, _vp, _i, ctypes.POINTER(ctypes.c_double), _i, _vp, _vp, _sz, _vp]),
    "nesti_patches_ref_max_ball": (_i, []),
    "nesti_patches_query_ref": (_i, [_cfgp, _vp, _i, _vp, _i, ctypes.POINTER(ctypes.c_double), _i, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp, _sz, _vp]),
    "nesti_model_describe": (_i, [_cfgp, ctypes.POINTER(_i), ctypes.POINTER(CTensor), _i]),
    "nesti_model_create": (_i, [_cfgp, ctypes.POINTER(CTensor), _i, _i, ctypes.POINTER(_vp)]),
    "nesti_model_destroy": (None, [_vp]),
    "nesti_model_set_gate_margin": (_i, [_vp, ctypes.c_float]),
    "nesti_model_cascade_stats": (_i, [_vp, ctypes.POINTER(CCascadeStats), _i, _vp]),
    "nesti_model_gate_error_export": (_i, [_vp, _vp, _vp]),
    "nesti_model_gate_error_import": (_i, [_vp, _vp, _i, _vp]),
    "nesti_experiment_mix_enable": (_i, [_i]),
    "nesti_model_set_expert_mix": (_i, [_vp, _i]),
    "nesti_model_set_gate_mix": (_i, [_vp, _i]),
    "nesti_model_set_x8_layers": (_i, [_vp, _i]),
    "nesti_model_set_x8_format": (_i, [_vp, _i]),
    "nesti_model_set_x8_guard": (_i, [_vp, ctypes.c_float]),
    "nesti_model_x8_guard_stats": (_i, [_vp, ctypes.POINTER(CX8GuardStats), _i, _vp]),
    "nesti_tower_workspace_bytes": (_sz, [_cfgp, _i, _i, _i]),
    "nesti_workspace_bytes": (_sz, [_vp, _i]),
    "nesti_model_mups_cstride": (_i, [_vp]),
    "nesti_model_mups_rows": (_i, [_vp]),
    "nesti_model_mups": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "nesti_gate_forward": (_i, [_vp, _vp, _i, _vp, _sz, _vp, _vp, _vp]),
    "nesti_experts_forward": (_i, [_vp, _vp, _vp, _i, _vp, _sz, _vp, _vp]),
    "nesti_forward": (_i, [_vp, _vp, _vp, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "nesti_estimate_workspace_bytes": (_sz, [_vp, _i]),
    "nesti_estimate_workspace_bytes_for_config": (_sz, [_cfgp, _i, _i]),
    "nesti_estimate_normals": (_i, [_vp, _vp, _i, _vp, _i, ctypes.POINTER(ctypes.c_double), _u64, _i, _i, _i, _vp, _sz,
                                    _vp, _sz, _vp, _vp, _vp, _vp]),
    "nesti_crc32c": (ctypes.c_uint32, [_vp, _sz, ctypes.c_uint32]),
    "nesti_f32_to_e2m3": (_i, [ctypes.c_float, ctypes.c_float]),
    "nesti_write_text_f32": (_i, [ctypes.c_char_p, _vp, ctypes.c_int64, _i]),
    "nesti_write_text_i32": (_i, [ctypes.c_char_p, _vp, ctypes.c_int64]),
    "nesti_estimate_normals_multi": (_i, [_vp, ctypes.POINTER(CShapeQueries), _i, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "nesti_refstream_create": (_vp, [ctypes.c_uint32]),
    "nesti_refstream_destroy": (None, [_vp]),
    "nesti_refstream_picks": (_i, [_vp, _vp, ctypes.c_int64, _i, _vp, ctypes.c_int64, _vp, ctypes.POINTER(ctypes.c_int64)]),
    "nesti_profile_enable": (_i, [_i]),
    "nesti_profile_read": (_i, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong)]),
    "nesti_model_macs": (_i, [_vp, _i, _i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                              ctypes.POINTER(ctypes.c_double)]),
}
PROF_CATEGORIES = ("conv8_k5", "conv8_k3", "taps_4_2", "one_by_one_fc", "mups", "pool", "patches")   # NESTI_PROF_*
PROF_CONV = PROF_CATEGORIES[:4]
PROF_PHASES = ("input", "gate", "recheck", "experts", "guard")                                      # NESTI_PHASE_*


def profile_read(lib):
    """``nesti_profile_read`` -> ({phase: {category: ms}}, {phase: {category: launches}})."""
    n = len(PROF_PHASES) * len(PROF_CATEGORIES)
    ms, nl = (ctypes.c_double * n)(), (ctypes.c_longlong * n)()
    lib.nesti_profile_read(ms, nl)
    k = len(PROF_CATEGORIES)
    return ({ph: {c: ms[i * k + j] for j, c in enumerate(PROF_CATEGORIES)} for i, ph in enumerate(PROF_PHASES)},
            {ph: {c: int(nl[i * k + j]) for j, c in enumerate(PROF_CATEGORIES)} for i, ph in enumerate(PROF_PHASES)})

_lib = None


def load():
    """Load the shared library once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64; it must be the HIP runtime of the process, so import torch BEFORE
    # dlopen()ing our library (otherwise /opt/rocm's copy is loaded first and the two runtimes disagree about devices)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NestiError(
            "libnesti_hip.so not found at %s -- build it first: "
            "python -c 'import __graft_entry__ as g; g.build()' (or make -C nesti-net_amd/csrc). "
            "There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    ver = (lib.nesti_version() or b"").decode()
    if "TIMING-EXPERIMENTS" in ver and os.environ.get("NESTI_ALLOW_TIMING_BUILD") != "1":
        raise NestiError("%s is a timing-only build (%s): it computes wrong results on purpose; set NESTI_ALLOW_TIMING_BUILD=1 "
                         "to load it for a measurement" % (LIB_PATH, ver))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().nesti_last_error()
        raise NestiError("%s failed: %s" % (what or "libnesti_hip call", msg.decode() if msg else "?"))


def ptr(t):
    """Device/host pointer of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return ctypes.c_void_p(t.data_ptr())
    return ctypes.c_void_p(t.ctypes.data)


def stream_ptr(stream=None):
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)
