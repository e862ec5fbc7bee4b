"""Format-level writer of a TF1 "tensor bundle" (LevelDB table + BundleEntryProto) and of the py2-style pickles a
trained reference run leaves in --results_path (test infrastructure shared by test_tf_ckpt.py and the GPU end-to-end test).
No TensorFlow-written file is available, so this pins the reader to the published formats, not to TF itself."""
import argparse
import json
import pickle
import struct
import sys
import types

import numpy as np


def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7f
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


_CRC_TABLE = []


def crc32c_py(data, crc=0):
    """Bitwise-table CRC-32C in pure Python: independent of the library's nesti_crc32c (which the reader uses)."""
    if not _CRC_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _CRC_TABLE.append(c)
    c = crc ^ 0xffffffff
    for b in bytes(data):
        c = _CRC_TABLE[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def masked(crc):
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def _crc_fn(kind):
    """'python': the independent implementation above (small bundles); 'native': the library's (the 100 MB model of the
    GPU end-to-end test); None: write zeros = "no checksum"."""
    if kind == "python":
        return crc32c_py
    if kind == "native":
        import nesti_net_amd  # noqa: F401
        from nesti_net_amd import tf_ckpt
        return tf_ckpt.crc32c
    return None


def _entry(dtype, shape, offset, size, crc=0):
    dims = b"".join(b"\x12" + _vi(len(_vi(d)) + 1) + b"\x08" + _vi(d) for d in shape)      # dim { size }
    msg = b"\x08" + _vi(dtype) + b"\x12" + _vi(len(dims)) + dims
    msg += b"\x18" + _vi(0) + b"\x20" + _vi(offset) + b"\x28" + _vi(size) + b"\x35" + struct.pack("<I", crc)
    return msg


def _block(items, restart_interval=16):
    buf, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(buf))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        buf += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        prev = k
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def write_bundle(prefix, tensors, per_block=7, crc="python"):
    """Minimal tensor-bundle writer: uncompressed table, several data blocks, one shard; masked crc32c per tensor and
    per table block like TF writes them (``crc``: see _crc_fn)."""
    fn = _crc_fn(crc)
    trailer = (lambda blk: b"\x00" + struct.pack("<I", masked(fn(blk + b"\x00")))) if fn else (lambda blk: b"\x00" + struct.pack("<I", 0))
    data, items = bytearray(), [(b"", b"\x08\x01")]                       # header entry (empty key)
    for name in sorted(tensors):
        a = np.ascontiguousarray(tensors[name], dtype=np.float32)
        items.append((name.encode(), _entry(1, a.shape, len(data), a.nbytes, masked(fn(a.tobytes())) if fn else 0)))
        data += a.tobytes()
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    out, index_items = bytearray(), []
    for i in range(0, len(items), per_block):
        blk = _block(items[i:i + per_block], restart_interval=3)
        index_items.append((items[min(i + per_block, len(items)) - 1][0], _vi(len(out)) + _vi(len(blk))))
        out += blk + trailer(blk)                                           # type + crc trailer
    meta = _block([])
    meta_h = _vi(len(out)) + _vi(len(meta))
    out += meta + trailer(meta)
    idx = _block(index_items, restart_interval=1)
    idx_h = _vi(len(out)) + _vi(len(idx))
    out += idx + trailer(idx)
    footer = (meta_h + idx_h).ljust(40, b"\x00") + struct.pack("<Q", 0xdb4775248b80fb57)
    open(prefix + ".index", "wb").write(bytes(out) + footer)


def tf_names(W, uniquified=False):
    """Rename this package's variables the way TF1 names them: the EMA shadows of the batch statistics live under the
    moments ops, re-entering the layer's scope (utils/tf_util.py:477-479).  ``uniquified``: the spelling TF produces when
    the name scope is re-entered a second time (``bn_1``) and the shadow variable itself was uniquified
    (``ExponentialMovingAverage_1``) -- every other layer, so that both spellings appear in one checkpoint."""
    out = {}
    flip = {}
    for k, v in W.items():
        if k.endswith("/bn/mean") or k.endswith("/bn/var"):
            sc = k[:k.rindex("/")]
            u = uniquified and flip.setdefault(sc, len(flip) % 2 == 0)
            op = "Squeeze" if k.endswith("/mean") else "Squeeze_1"
            out["%s/%s%s/moments/%s/ExponentialMovingAverage%s" % (sc, sc, "_1" if u else "", op, "_1" if u else "")] = v
        else:
            out[k] = v
    out["beta1_power"] = np.float32(0.5)                                    # optimizer slots are ignored
    return out


def write_model_dir(path, cfg, W, uniquified=False, per_block=50, crc="native"):
    """parameters.p + gmm.p + model.ckpt.* as train_n_est_w_experts.py:120-122, 248-250, 353-354 leave them."""
    import os
    ns = argparse.Namespace(patch_radius=list(cfg.patch_radius), num_point=cfg.num_point, n_experts=cfg.n_experts,
                            num_gaussians=cfg.n_gaussians, gmm_variance=cfg.gmm_variance, model="experts_n_est",
                            expert_loss_type="simple", loss_type="cos",
                            expert_dict=json.dumps({str(k): json.dumps(v) for k, v in cfg.expert_dict.items()}))
    pickle.dump(ns, open(os.path.join(path, "parameters.p"), "wb"), protocol=2)
    mod = types.ModuleType("sklearn.mixture.gaussian_mixture")
    cls = type("GaussianMixture", (object,), {"__module__": "sklearn.mixture.gaussian_mixture"})
    mod.GaussianMixture = cls
    sys.modules["sklearn.mixture.gaussian_mixture"] = mod
    try:
        g = cls()
        G = cfg.n_gaussians ** 3
        g.weights_ = np.ones(G) / G
        g.means_ = np.zeros((G, 3))
        g.covariances_ = cfg.gmm_variance * np.ones((G, 3))
        pickle.dump(g, open(os.path.join(path, "gmm.p"), "wb"), protocol=2)
    finally:
        del sys.modules["sklearn.mixture.gaussian_mixture"]
    write_bundle(os.path.join(path, "model.ckpt"), tf_names(W, uniquified), per_block=per_block, crc=crc)


