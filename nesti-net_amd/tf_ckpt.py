"""Reader for the artefacts a trained reference run leaves in ``--results_path``
(``train_n_est_w_experts.py:120-122, 248-250, 353-354``; consumed at
``test_n_est_w_experts.py:46-54, 98-105, 201``):

* ``model.ckpt.index`` + ``model.ckpt.data-00000-of-00001`` -- a TensorFlow "tensor bundle": the
  index is a LevelDB-format table (prefix-compressed keys, restart arrays, 48-byte footer) whose
  values are ``BundleEntryProto`` messages (dtype, shape, shard_id, offset, size, crc32c) pointing
  into the data shard;
* ``parameters.p`` -- the pickled argparse Namespace of the training run (Python 2);
* ``gmm.p`` -- the pickled sklearn GaussianMixture of the Gaussian grid (Python 2).

No TensorFlow, sklearn or Python 2 is needed.  TensorFlow itself is not installable here and the
reference ships no checkpoint, so no TF-written file exists to read; the format primitives are pinned by independent
known answers instead (``tests/test_tf_ckpt.py``: the RFC 3720 / LevelDB CRC-32C vectors and the masked form, varint
edge cases, a hand-assembled prefix-compressed block, the table magic), and whole bundles against the format-level
writer in ``tests/ckpt_writer.py``.  Like ``saver.restore``, :func:`read_bundle` verifies the stored checksums (table
blocks and tensors) and refuses a file whose bytes do not match them.
"""
import io
import json
import os
import pickle
import re
import struct

import numpy as np

from .config import NestiConfig

TABLE_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 19: np.float16}   # tensorflow/core/framework/types.proto


# ---- protobuf / varint helpers -----------------------------------------------------------------
def _varint(buf, pos):
    res, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        res |= (b & 0x7f) << shift
        if not b & 0x80:
            return res, pos
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) of one protobuf message."""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fn, wt, v


def _parse_entry(buf):
    """BundleEntryProto (tensorflow/core/protobuf/tensor_bundle.proto)."""
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0}
    for fn, wt, v in _fields(buf):
        if fn == 1:
            e["dtype"] = v
        elif fn == 2:                                  # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
            for f2, _, v2 in _fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _fields(v2):
                        if f3 == 1:
                            size = v3
                    e["shape"].append(size)
        elif fn == 3:
            e["shard_id"] = v
        elif fn == 4:
            e["offset"] = v
        elif fn == 5:
            e["size"] = v
        elif fn == 6 and wt == 5:                      # fixed32 crc32c (masked) of the tensor's bytes
            e["crc32c"] = struct.unpack("<I", v)[0]
    return e


# ---- checksums (tensorflow/core/lib/hash/crc32c.h) ---------------------------------------------
_MASK_DELTA = 0xa282ead8


_CRC_TABLES = None


def _crc_tables():
    """Slicing-by-8 tables of the Castagnoli polynomial (reflected 0x82F63B78), built once with numpy."""
    global _CRC_TABLES
    if _CRC_TABLES is None:
        t0 = np.arange(256, dtype=np.uint32)
        for _ in range(8):
            t0 = np.where(t0 & 1, (t0 >> 1) ^ np.uint32(0x82F63B78), t0 >> 1).astype(np.uint32)
        tabs = [t0]
        for _ in range(7):
            prev = tabs[-1]
            tabs.append((t0[prev & 0xff] ^ (prev >> 8)).astype(np.uint32))
        _CRC_TABLES = [t.tolist() for t in tabs]
    return _CRC_TABLES


def crc32c_py(data, crc=0):
    """CRC-32C without the native library (slicing-by-8 over table lookups; ~10 MB/s -- fine for index blocks, slow for the
    ~400 MB of tensor data of a full checkpoint)."""
    t = _crc_tables()
    buf = bytes(data) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).tobytes()
    c = (~crc) & 0xffffffff
    n8 = len(buf) // 8 * 8
    t0, t1, t2, t3, t4, t5, t6, t7 = t
    for lo, hi in struct.iter_unpack("<II", buf[:n8]):
        lo ^= c
        c = (t7[lo & 0xff] ^ t6[(lo >> 8) & 0xff] ^ t5[(lo >> 16) & 0xff] ^ t4[lo >> 24] ^
             t3[hi & 0xff] ^ t2[(hi >> 8) & 0xff] ^ t1[(hi >> 16) & 0xff] ^ t0[hi >> 24])
    for b in buf[n8:]:
        c = t0[(c ^ b) & 0xff] ^ (c >> 8)
    return (~c) & 0xffffffff


def crc32c(data, crc=0):
    """CRC-32C of bytes / a contiguous uint8 array: ``nesti_crc32c`` of libnesti_hip.so (host code) when the library can be
    loaded, the pure-Python routine otherwise -- reading a checkpoint's index must not depend on the ROCm runtime."""
    from . import _lib
    try:
        lib = _lib.load()
    except (OSError, _lib.NestiError):
        return crc32c_py(data, crc)
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    return int(lib.nesti_crc32c(_lib.ptr(buf) if buf.size else None, buf.size, crc))


def mask_crc(crc):
    """The stored form of a checksum: rotate right by 15 and add a constant (crc32c::Mask)."""
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
    rot = (masked - _MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ---- LevelDB table ---------------------------------------------------------------------------
def _read_block(data, offset, size, verify=True):
    raw = data[offset:offset + size]
    ctype = data[offset + size]                        # 1-byte compression type + 4-byte masked crc32c(block + type) follow
    if ctype != 0:
        raise ValueError("compressed table blocks (type %d) are not supported" % ctype)
    stored = struct.unpack("<I", data[offset + size + 1:offset + size + 5])[0]
    # a table block ALWAYS carries its trailer (leveldb's BlockBuilder / TF's table_builder.cc write it unconditionally), so a
    # zeroed field is a corrupt block, not an absent checksum (unlike the optional crc32c field of a BundleEntryProto)
    if verify and unmask_crc(stored) != crc32c(data[offset:offset + size + 1]):
        raise ValueError("table block at offset %d fails its crc32c" % offset)
    n_restarts = struct.unpack("<I", raw[-4:])[0]
    end = len(raw) - 4 - 4 * n_restarts
    pos, key = 0, b""
    out = []
    while pos < end:
        shared, pos = _varint(raw, pos)
        non_shared, pos = _varint(raw, pos)
        vlen, pos = _varint(raw, pos)
        key = key[:shared] + raw[pos:pos + non_shared]
        pos += non_shared
        out.append((key, raw[pos:pos + vlen]))
        pos += vlen
    return out


def read_index(index_path):
    """-> {variable name: BundleEntryProto dict} (the header entry with the empty key is dropped)."""
    return read_index_and_header(index_path)[0]


def read_index_and_header(index_path):
    """-> ({variable name: BundleEntryProto dict}, {'num_shards': n}) -- the BundleHeaderProto sits on the empty key
    (tensorflow/core/protobuf/tensor_bundle.proto: num_shards = 1, endianness = 2, version = 3)."""
    data = open(index_path, "rb").read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow tensor-bundle index" % index_path)
    footer = data[-48:]
    _, p = _varint(footer, 0)                          # metaindex handle (unused)
    _, p = _varint(footer, p)
    idx_off, p = _varint(footer, p)
    idx_size, p = _varint(footer, p)
    entries, header = {}, {"num_shards": 0}
    for _, handle in _read_block(data, idx_off, idx_size):
        off, q = _varint(handle, 0)
        size, q = _varint(handle, q)
        for key, val in _read_block(data, off, size):
            if key:
                entries[key.decode()] = _parse_entry(val)
            else:
                for fn, wt, v in _fields(val):
                    if fn == 1 and wt == 0:
                        header["num_shards"] = int(v)
                    elif fn == 2 and wt == 0 and v != 0:
                        raise ValueError("%s was written on a big-endian host" % index_path)
    if header["num_shards"] <= 0:          # no header entry (not a TF-written bundle): fall back to the entries' shard ids
        header["num_shards"] = 1 + max([e["shard_id"] for e in entries.values()] + [0])
    return entries, header


def read_bundle(prefix, verify=True):
    """``saver.restore`` without TF: prefix e.g. ``.../model.ckpt`` -> {name: ndarray}.  ``verify``: check every tensor's
    stored crc32c like TF does on restore (an all-zero field means "not written" and is skipped)."""
    entries, header = read_index_and_header(prefix + ".index")
    n = header["num_shards"]                           # the data files are named after the HEADER's shard count (naming.cc)
    shards = {}
    out = {}
    for name, e in entries.items():
        if e["dtype"] not in _DTYPES:
            continue                                   # e.g. string tensors of the saver itself
        sid = e["shard_id"]
        if sid >= n:
            raise ValueError("tensor %s lives in shard %d but the bundle header declares %d shard(s)" % (name, sid, n))
        if e["size"] == 0:
            out[name] = np.zeros(e["shape"], _DTYPES[e["dtype"]])
            continue
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, n), dtype=np.uint8, mode="r")
        raw = np.asarray(shards[sid][e["offset"]:e["offset"] + e["size"]])
        if verify and e["crc32c"] != 0 and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise ValueError("tensor %s fails its crc32c (corrupt or truncated %s)" % (name, prefix))
        out[name] = raw.view(_DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
    return out


# ---- variable name mapping ----------------------------------------------------------------------
def map_variables(raw, expected):
    """TF variable names -> this package's names (``weights.describe``).

    ``<scope>/weights|biases`` and ``<scope>/bn/beta|gamma`` are used verbatim
    (``utils/tf_util.py:292,301,473-476``).  The EMA shadows of the batch statistics are named by
    ``tf.train.ExponentialMovingAverage`` after the ``moments`` ops (``utils/tf_util.py:477-479``),
    so they are DISCOVERED in the checkpoint's index rather than spelled here: under ``<scope>/bn/`` the keys whose
    last component is ``ExponentialMovingAverage`` (or a uniquified ``ExponentialMovingAverage_<n>``); the op name in
    front of it tells the statistic -- ``tf.nn.moments`` squeezes the mean in ``Squeeze`` and the variance in
    ``Squeeze_1`` (or names them ``mean`` / ``variance``).  Exactly one of each must exist; anything else raises."""
    out = {}
    for name, shape in expected.items():
        if name in raw:
            arr = raw[name]
        elif name.endswith("/bn/mean") or name.endswith("/bn/var"):
            scope = name[:name.rindex("/")] + "/"
            cands = sorted(k for k in raw if k.startswith(scope) and re.search(r"/ExponentialMovingAverage(_\d+)?$", k))
            if len(cands) != 2:
                raise KeyError("expected 2 EMA shadow variables under %s, found %s" % (scope, cands))
            ops = [k.split("/")[-2] for k in cands]                  # the moments op the shadow belongs to
            is_var = [op in ("Squeeze_1", "variance") for op in ops]
            is_mean = [op in ("Squeeze", "mean") for op in ops]
            if sum(is_var) != 1 or sum(is_mean) != 1:
                raise KeyError("cannot tell mean from variance among %s" % cands)
            arr = raw[cands[is_var.index(name.endswith("/bn/var"))]]
        else:
            raise KeyError("variable %s not found in the checkpoint" % name)
        if tuple(arr.shape) != tuple(shape):
            raise ValueError("%s: checkpoint shape %s != graph shape %s" % (name, tuple(arr.shape), tuple(shape)))
        out[name] = np.ascontiguousarray(arr, dtype=np.float32)
    return out


# ---- parameters.p / gmm.p -------------------------------------------------------------------------
class _Stub:
    """Stand-in for classes that do not exist here (sklearn.mixture.gaussian_mixture.GaussianMixture)."""

    def __setstate__(self, state):
        self.__dict__.update(state)


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("sklearn"):
            return _Stub
        if module == "copy_reg":
            module = "copyreg"
        if module == "__builtin__":
            module = "builtins"
        return super().find_class(module, name)


def _load_py2_pickle(path):
    with open(path, "rb") as f:
        return _Unpickler(io.BytesIO(f.read()), encoding="latin1").load()


def load_parameters(path):
    """``parameters.p`` (the pickled argparse namespace of the training script) -> NestiConfig
    (``test_n_est_w_experts.py:46-54``, ``test_n_est.py:35-42``, ``test_n_est_w_switching.py:31-37``)."""
    ns = _load_py2_pickle(path)
    variance = getattr(ns, "gmm_variance", 0.0156)
    n_gauss = getattr(ns, "num_gaussians", getattr(ns, "n_gaussians", 8))   # train_*.py: --num_gaussians
    model = getattr(ns, "model", "experts_n_est")
    radius = [float(r) for r in ns.patch_radius]
    common = dict(patch_radius=radius, num_point=int(ns.num_point), n_gaussians=int(n_gauss), gmm_variance=float(variance))
    if model == "experts_n_est":
        ed = json.loads(ns.expert_dict)                                      # JSON-in-JSON, :53-54
        ed = {int(k): (json.loads(v) if isinstance(v, str) else v) for k, v in ed.items()}
        return NestiConfig(n_experts=int(ns.n_experts), expert_dict=ed, **common)
    base = NestiConfig.for_model(model)                                      # raises on an unknown model name
    if model == "ss_norm_est" and len(radius) != 1:
        raise ValueError("ss_norm_est takes one patch radius, parameters.p has %s" % radius)
    ed = {0: list(range(len(radius)))} if base.n_experts == 1 else base.expert_dict
    return NestiConfig(n_experts=base.n_experts, expert_dict=ed, arch=base.arch, **common)


def load_gmm(path):
    """``gmm.p`` -> (weights_, means_, covariances_) as float64 arrays."""
    g = _load_py2_pickle(path)
    return np.asarray(g.weights_), np.asarray(g.means_), np.asarray(g.covariances_)


def load_reference_model(results_path):
    """Everything ``test_n_est_w_experts.py`` reads from a trained-model directory ->
    (NestiConfig, weights dict) ready for :class:`NestiNet` / ``weights.save``."""
    from . import weights as wts
    cfg = load_parameters(os.path.join(results_path, "parameters.p"))
    gp = os.path.join(results_path, "gmm.p")
    if os.path.exists(gp):
        w, mu, cov = load_gmm(gp)
        n = int(round(len(w) ** (1.0 / 3.0)))
        if n != cfg.n_gaussians or abs(float(cov.flat[0]) - cfg.gmm_variance) > 1e-12:
            cfg.n_gaussians, cfg.gmm_variance = n, float(cov.flat[0])
    raw = read_bundle(os.path.join(results_path, "model.ckpt"))
    return cfg, map_variables(raw, wts.describe(cfg))
