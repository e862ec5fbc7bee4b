"""Model variables: enumeration (from the C++ graph builder), seeded synthetic
initialisation, and an on-disk container.

The reference restores a TF1 checkpoint (``test_n_est_w_experts.py:98-105``); no
checkpoint is available (reference ``.gitignore:1-2``), so weights are synthetic:
Xavier-uniform like ``utils/tf_util.py:46-47``, with non-trivial batch-norm
statistics so that the BN fold is exercised.  Variable names follow the
reference's scopes (``<scope>/weights``, ``<scope>/biases``, ``<scope>/bn/beta|gamma``
-- ``utils/tf_util.py:292,301,473-476``); the EMA shadows are called
``<scope>/bn/mean`` and ``<scope>/bn/var`` here.
"""
import ctypes
import os
import struct
from collections import OrderedDict

import numpy as np

from . import _lib
from .config import NestiConfig

WEIGHT_SEED = 20190615
MAGIC = b"NSTW1\0\0\0"
BN_STATS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "synth_bn_stats.npz")


def describe(cfg: NestiConfig):
    """Ordered {name: shape} of every variable the graph for ``cfg`` expects
    (``nesti_model_describe``)."""
    lib = _lib.load()
    c = cfg.to_c()
    n = ctypes.c_int(0)
    _lib.check(lib.nesti_model_describe(ctypes.byref(c), ctypes.byref(n), None, 0), "nesti_model_describe")
    infos = (_lib.CTensor * n.value)()
    _lib.check(lib.nesti_model_describe(ctypes.byref(c), ctypes.byref(n), infos, n.value), "nesti_model_describe")
    out = OrderedDict()
    for t in infos:
        out[t.name.decode()] = tuple(int(t.dims[d]) for d in range(t.ndim))
    return out


def synthetic_weights(cfg: NestiConfig, seed=WEIGHT_SEED, bn="auto"):
    """Deterministic float32 variables for ``cfg``.

    ``bn``: 'random' -- arbitrary batch-norm EMA statistics (exercises the fold, but the activations of the deep ReLU
    stack are then dominated by their input-independent mean); 'calibrated' -- the EMA statistics a training run would
    leave behind: per-channel mean / variance of every layer's pre-activation over a sample of real MuPS tensors
    (``scripts/make_synth_bn_stats.py`` -> ``data/synth_bn_stats.npz``, generated for the default configuration and
    ``WEIGHT_SEED``); 'auto' -- calibrated when the committed statistics fit this configuration, else random."""
    rng = np.random.RandomState(seed)
    W = OrderedDict()
    for name, shape in describe(cfg).items():
        if name.endswith("/weights"):
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
            # He-style gain keeps activations O(1) through the ReLU stack; uniform like xavier_initializer
            lim = np.sqrt(6.0 / fan_in)
            W[name] = rng.uniform(-lim, lim, size=shape).astype(np.float32)
        elif name.endswith("/biases"):
            W[name] = rng.uniform(-0.1, 0.1, size=shape).astype(np.float32)
        elif name.endswith("/bn/beta"):
            W[name] = rng.uniform(-0.1, 0.1, size=shape).astype(np.float32)
        elif name.endswith("/bn/gamma"):
            W[name] = rng.uniform(0.8, 1.2, size=shape).astype(np.float32)
        elif name.endswith("/bn/mean"):
            W[name] = rng.uniform(-0.2, 0.2, size=shape).astype(np.float32)
        elif name.endswith("/bn/var"):
            W[name] = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        else:
            raise ValueError("unexpected variable %s" % name)
    # spread the gate's arg-max over the experts: fc4 has ReLU before the softmax
    # (models/experts_n_est.py:174), so give it O(1) positive biases
    if "fc4noise/biases" in W:
        W["fc4noise/biases"] = rng.uniform(0.5, 1.5, size=W["fc4noise/biases"].shape).astype(np.float32)
        W["fc4noise/weights"] = (W["fc4noise/weights"] * 4.0).astype(np.float32)
    if bn not in ("auto", "random", "calibrated"):
        raise ValueError("bn must be 'auto', 'random' or 'calibrated'")
    if bn != "random":
        ok = os.path.exists(BN_STATS_FILE) and seed == WEIGHT_SEED
        stats = dict(np.load(BN_STATS_FILE)) if ok else {}
        ok = ok and int(stats.pop("seed")) == seed
        names = [n for n in W if n.endswith("/bn/mean") or n.endswith("/bn/var")]
        ok = ok and all(n in stats and stats[n].shape == W[n].shape for n in names) and len(names) == len(stats)
        if ok:
            for n in names:
                W[n] = stats[n].astype(np.float32)
        elif bn == "calibrated":
            raise ValueError("no committed batch-norm statistics for this configuration / seed (scripts/make_synth_bn_stats.py)")
    return W


def save(path, W, cfg: NestiConfig = None):
    """Write variables (+ optional config JSON) to one file."""
    with open(path, "wb") as f:
        f.write(MAGIC)
        cj = (cfg.to_json() if cfg is not None else "").encode()
        f.write(struct.pack("<II", len(W), len(cj)))
        f.write(cj)
        for name, a in W.items():
            a = np.ascontiguousarray(a, dtype=np.float32)
            nb = name.encode()
            f.write(struct.pack("<II", len(nb), a.ndim))
            f.write(nb)
            f.write(struct.pack("<%dq" % a.ndim, *a.shape))
            f.write(a.tobytes())


def load(path):
    """Inverse of :func:`save` -> (weights dict, NestiConfig or None)."""
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("%s is not an NSTW1 weight file" % path)
        n, cl = struct.unpack("<II", f.read(8))
        cj = f.read(cl).decode()
        W = OrderedDict()
        for _ in range(n):
            ln, nd = struct.unpack("<II", f.read(8))
            name = f.read(ln).decode()
            shape = struct.unpack("<%dq" % nd, f.read(8 * nd))
            cnt = int(np.prod(shape)) if nd else 1
            W[name] = np.frombuffer(f.read(4 * cnt), dtype=np.float32).reshape(shape).copy()
    return W, (NestiConfig.from_json(cj) if cj else None)
