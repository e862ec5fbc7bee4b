"""TF1 checkpoint / py2 pickle readers (N1) against files written here at the FORMAT level
(LevelDB table + BundleEntryProto encoders below).  No TensorFlow-written file is available, so
this pins the reader to the published formats, not to TF itself."""
import argparse
import json
import os
import pickle
import struct
import sys
import types

import numpy as np
import pytest

from ckpt_writer import crc32c_py, masked, tf_names, write_bundle


def test_bundle_roundtrip_and_name_mapping(tmp_path):
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import tf_ckpt
    rng = np.random.RandomState(0)
    expected = {"a_conv1/weights": (1, 1, 1, 4, 6), "a_conv1/biases": (6,), "a_conv1/bn/beta": (6,),
                "a_conv1/bn/gamma": (6,), "a_conv1/bn/mean": (6,), "a_conv1/bn/var": (6,),
                "fc4x/weights": (5, 3), "fc4x/biases": (3,)}
    W = {k: rng.randn(*s).astype(np.float32) for k, s in expected.items()}
    for i in range(40):                                                      # force several table blocks
        W["pad%03d/weights" % i] = rng.randn(2, 2).astype(np.float32)
        expected["pad%03d/weights" % i] = (2, 2)
    prefix = str(tmp_path / "model.ckpt")
    write_bundle(prefix, tf_names(W))
    raw = tf_ckpt.read_bundle(prefix)
    assert "beta1_power" in raw and len(raw) == len(W) + 1
    got = tf_ckpt.map_variables(raw, expected)
    assert set(got) == set(expected)
    for k in expected:
        assert np.array_equal(got[k], W[k]), k
    # the spelling with re-entered name scopes: <scope>/bn/<scope>/bn_1/moments/Squeeze_1/ExponentialMovingAverage_1
    prefix_u = str(tmp_path / "model_u.ckpt")
    named = tf_names(W, uniquified=True)
    assert any(k.endswith("ExponentialMovingAverage_1") and "/bn_1/" in k for k in named)
    write_bundle(prefix_u, named)
    got_u = tf_ckpt.map_variables(tf_ckpt.read_bundle(prefix_u), expected)
    assert all(np.array_equal(got_u[k], W[k]) for k in expected)
    # two shadow variables of the same kind under one scope cannot be told apart: an error, not a guess
    dup = dict(raw)
    dup["a_conv1/bn/a_conv1/bn_1/moments/Squeeze/ExponentialMovingAverage"] = raw["a_conv1/bn/a_conv1/bn/moments/Squeeze/ExponentialMovingAverage"]
    with pytest.raises(KeyError):
        tf_ckpt.map_variables(dup, expected)
    with pytest.raises(KeyError):
        tf_ckpt.map_variables(raw, {"missing/weights": (1,)})
    with pytest.raises(ValueError):
        tf_ckpt.map_variables(raw, {"fc4x/weights": (3, 5)})
    with pytest.raises(ValueError):
        open(prefix + ".index", "ab").write(b"x")
        tf_ckpt.read_index(prefix + ".index")


def test_py2_pickles_and_full_model_dir(tmp_path, monkeypatch):
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import tf_ckpt, weights
    from nesti_net_amd.config import NestiConfig
    cfg = NestiConfig()
    ns = argparse.Namespace(patch_radius=[0.01, 0.03, 0.05], num_point=512, n_experts=7, num_gaussians=8, gmm_variance=0.0156,
                            model="experts_n_est",
                            expert_loss_type="simple", loss_type="cos",
                            expert_dict=json.dumps({str(k): json.dumps(v) for k, v in cfg.expert_dict.items()}))
    pickle.dump(ns, open(str(tmp_path / "parameters.p"), "wb"), protocol=2)
    mod = types.ModuleType("sklearn.mixture.gaussian_mixture")
    cls = type("GaussianMixture", (object,), {"__module__": "sklearn.mixture.gaussian_mixture"})
    mod.GaussianMixture = cls
    sys.modules["sklearn.mixture.gaussian_mixture"] = mod
    try:
        g = cls()
        g.weights_ = np.ones(512) / 512
        g.means_ = np.zeros((512, 3))
        g.covariances_ = 0.0156 * np.ones((512, 3))
        pickle.dump(g, open(str(tmp_path / "gmm.p"), "wb"), protocol=2)
    finally:
        del sys.modules["sklearn.mixture.gaussian_mixture"]
    cfg2 = tf_ckpt.load_parameters(str(tmp_path / "parameters.p"))
    assert cfg2 == cfg
    # the ablation drivers pickle the same kind of namespace, without expert fields
    # (train_n_est.py:99, train_n_est_w_switching.py:111); --model selects the graph
    for model, radius in (("ss_norm_est", [0.05]), ("ms_norm_est", [0.01, 0.03, 0.05]), ("ms_sw_n_est", [0.01, 0.05])):
        ns_a = argparse.Namespace(patch_radius=radius, num_point=512, num_gaussians=8, gmm_variance=0.0156, model=model)
        pickle.dump(ns_a, open(str(tmp_path / "parameters_a.p"), "wb"), protocol=2)
        got = tf_ckpt.load_parameters(str(tmp_path / "parameters_a.p"))
        want = NestiConfig.for_model(model)
        assert got.arch == want.arch and got.patch_radius == radius and got.n_towers == want.n_towers
        weights.describe(got)          # the graph builder accepts it
    w, mu, cov = tf_ckpt.load_gmm(str(tmp_path / "gmm.p"))
    assert w.shape == (512,) and cov[0, 0] == 0.0156
    # a whole trained-model directory; the real graph's variable list restricted to its small tensors
    # (the conv weights alone are 700 MB) so the CPU suite stays light
    full = weights.describe(cfg)
    assert len(full) == 976
    exp = {k: v for k, v in full.items() if int(np.prod(v)) <= 1 << 17}
    assert len(exp) > 700 and "fc4noise/weights" in exp and "inception1gating_conv_conv1/bn/mean" in exp
    monkeypatch.setattr(weights, "describe", lambda c: exp)
    rng = np.random.RandomState(1)
    W = {k: rng.rand(*s).astype(np.float32) for k, s in exp.items()}
    write_bundle(str(tmp_path / "model.ckpt"), tf_names(W), per_block=50)
    cfg3, W3 = tf_ckpt.load_reference_model(str(tmp_path) + os.sep)
    assert cfg3 == cfg and list(W3) == list(exp)
    assert all(np.array_equal(W3[k], W[k]) for k in exp)


# ---- format primitives against independent known answers (no writer of ours involved) ----------------------------------
def test_crc32c_known_answers_and_masked_form():
    """CRC-32C test vectors of RFC 3720 B.4 (the same ones LevelDB's crc32c_test.cc and TensorFlow's crc32c_test.cc hold),
    the classic check value, chaining, and crc32c::Mask / Unmask with its published constant."""
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import tf_ckpt
    iscsi_read = bytes([0x01, 0xc0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x14, 0, 0, 0, 0, 0, 0x04, 0, 0, 0, 0, 0x14,
                        0, 0, 0, 0x18, 0x28, 0, 0, 0, 0, 0, 0, 0, 0x02, 0, 0, 0, 0, 0, 0, 0])
    vectors = [(bytes(32), 0x8a9136aa), (b"\xff" * 32, 0x62a8ab43), (bytes(range(32)), 0x46dd794e),
               (bytes(range(31, -1, -1)), 0x113fdb5c), (iscsi_read, 0xd9963a56), (b"123456789", 0xe3069283), (b"", 0)]
    for data, want in vectors:
        assert tf_ckpt.crc32c(data) == want, data
        assert crc32c_py(data) == want                                     # the test writer's own implementation too
    # Extend: crc(a + b) == crc(b, crc(a)); unaligned starts and lengths that are not multiples of 8 (the slice-by-8 tails)
    blob = np.random.RandomState(3).bytes(1000)
    for cut in (0, 1, 7, 8, 9, 500, 999, 1000):
        assert tf_ckpt.crc32c(blob[cut:], tf_ckpt.crc32c(blob[:cut])) == crc32c_py(blob)
    arr = np.frombuffer(blob, np.uint8)
    assert tf_ckpt.crc32c(arr[3:997]) == crc32c_py(blob[3:997])
    # the reader's pure-Python fallback (used when libnesti_hip.so cannot be loaded): same vectors, chaining, odd tails
    for data, want in vectors:
        assert tf_ckpt.crc32c_py(data) == want
    assert tf_ckpt.crc32c_py(b"123456789") == 0xe3069283
    for cut in (0, 1, 7, 8, 9, 500, 999, 1000):
        assert tf_ckpt.crc32c_py(blob[cut:], tf_ckpt.crc32c_py(blob[:cut])) == crc32c_py(blob)
    # crc32c::Mask: rotate right 15, add 0xa282ead8 (mod 2^32)
    assert tf_ckpt.mask_crc(0) == 0xa282ead8 and tf_ckpt.mask_crc(0xe3069283) == 0xc78ab0e5 == masked(0xe3069283)
    for c in (0, 1, 0x8a9136aa, 0xffffffff, 0x5d7d1528):
        assert tf_ckpt.unmask_crc(tf_ckpt.mask_crc(c)) == c and tf_ckpt.mask_crc(tf_ckpt.mask_crc(c)) != c


def test_varint_and_protobuf_field_edge_cases():
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import tf_ckpt
    for raw, want in ((b"\x00", 0), (b"\x7f", 127), (b"\x80\x01", 128), (b"\xac\x02", 300), (b"\xff\xff\xff\xff\x0f", 2 ** 32 - 1),
                      (b"\x80\x80\x80\x80\x10", 2 ** 32), (b"\xff" * 9 + b"\x01", 2 ** 64 - 1)):
        assert tf_ckpt._varint(raw + b"\x55", 0) == (want, len(raw))
    # BundleEntryProto by hand: dtype DT_FLOAT, shape [3, 300], shard 0, offset 2^32, size 3600, crc fixed32
    msg = (b"\x08\x01" + b"\x12\x09" + b"\x12\x02\x08\x03" + b"\x12\x03\x08\xac\x02" + b"\x18\x00" +
           b"\x20\x80\x80\x80\x80\x10" + b"\x28\x90\x1c" + b"\x35\xe5\xb0\x8a\xc7")
    e = tf_ckpt._parse_entry(msg)
    assert e == {"dtype": 1, "shape": [3, 300], "shard_id": 0, "offset": 2 ** 32, "size": 3600, "crc32c": 0xc78ab0e5}


def test_prefix_compressed_block_and_footer_by_hand(tmp_path):
    """A LevelDB data block assembled byte by byte (leveldb/table/block_builder.cc: shared | non_shared | value_len |
    key delta | value, restart array, restart count), wrapped into the smallest table, read back through read_index."""
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import tf_ckpt
    val = b"\x08\x01\x12\x04\x12\x02\x08\x05\x28\x14"                # DT_FLOAT, shape [5], size 20
    entries = (b"\x00\x00\x02\x08\x01" +                               # restart 0: key "" (bundle header)
               b"\x00\x0b" + bytes([len(val)]) + b"conv1/biase" + val +   # "conv1/biase"
               b"\x0b\x01" + bytes([len(val)]) + b"s" + val +             # shares 11 bytes -> "conv1/biases"
               b"\x00\x0d" + bytes([len(val)]) + b"conv1/weights" + val + # restart 1 (offset below): full key again
               b"\x06\x03" + bytes([len(val)]) + b"x/y" + val)            # shares "conv1/" -> "conv1/x/y"
    r1 = 5 + (2 + 1 + 11 + len(val)) + (2 + 1 + 1 + len(val))
    block = entries + struct.pack("<III", 0, r1, 2)
    got = tf_ckpt._read_block(block + b"\x00" + struct.pack("<I", tf_ckpt.mask_crc(tf_ckpt.crc32c(block + b"\x00"))), 0, len(block))
    assert [k for k, _ in got] == [b"", b"conv1/biase", b"conv1/biases", b"conv1/weights", b"conv1/x/y"]
    assert all(v == val for _, v in got[1:])
    # a flipped byte must be caught by the block checksum
    bad = bytearray(block + b"\x00" + struct.pack("<I", tf_ckpt.mask_crc(tf_ckpt.crc32c(block + b"\x00"))))
    bad[20] ^= 0x40
    zeroed = block + b"\x00" + b"\x00\x00\x00\x00"                                # a table block always carries its trailer:
    with pytest.raises(ValueError, match="crc32c"):                                # an all-zero one is corruption, not "absent"
        tf_ckpt._read_block(zeroed, 0, len(block))
    with pytest.raises(ValueError, match="crc32c"):
        tf_ckpt._read_block(bytes(bad), 0, len(block))
    # the smallest table: data block, empty metaindex block, index block, 48-byte footer ending in the LevelDB magic
    def trailer(b):
        return b + b"\x00" + struct.pack("<I", tf_ckpt.mask_crc(tf_ckpt.crc32c(b + b"\x00")))
    meta = struct.pack("<II", 0, 1)
    handle = bytes([0]) + bytes([len(block)])                              # varint offset 0, varint size (< 128)
    assert len(block) < 128
    index = b"\x00\x01" + bytes([len(handle)]) + b"d" + handle + struct.pack("<II", 0, 1)
    body = trailer(block)
    meta_off = len(body)
    body += trailer(meta)
    idx_off = len(body)
    body += trailer(index)
    footer = (bytes([meta_off, len(meta)]) + bytes([idx_off, len(index)])).ljust(40, b"\x00") + bytes.fromhex("57fb808b247547db")
    path = str(tmp_path / "hand.index")
    open(path, "wb").write(body + footer)
    ents = tf_ckpt.read_index(path)
    assert sorted(ents) == ["conv1/biase", "conv1/biases", "conv1/weights", "conv1/x/y"]
    assert ents["conv1/weights"]["shape"] == [5] and ents["conv1/weights"]["size"] == 20
    open(path, "wb").write(body + footer[:-1] + b"\x00")
    with pytest.raises(ValueError, match="not a TensorFlow tensor-bundle index"):
        tf_ckpt.read_index(path)


def test_corrupt_tensor_bytes_are_refused(tmp_path):
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import tf_ckpt
    prefix = str(tmp_path / "m.ckpt")
    write_bundle(prefix, {"a/weights": np.arange(12, dtype=np.float32).reshape(3, 4), "a/biases": np.ones(4, np.float32)})
    assert np.array_equal(tf_ckpt.read_bundle(prefix)["a/weights"], np.arange(12, dtype=np.float32).reshape(3, 4))
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    raw[-3] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="fails its crc32c"):
        tf_ckpt.read_bundle(prefix)
    assert tf_ckpt.read_bundle(prefix, verify=False)["a/weights"].shape == (3, 4)


# ---- a whole TF-1.12-shaped checkpoint assembled from the published formats, without tests/ckpt_writer.py -------------------
# (VERDICT r04 item 8).  Sources: leveldb/doc/table_format.md + table/block_builder.cc (entries = varint shared | varint
# non_shared | varint value_len | key delta | value; restart array of fixed32 offsets + fixed32 count; block trailer = type byte
# + masked crc32c; footer = metaindex handle | index handle, padded to 40 bytes, + 8-byte magic), tensorflow/core/protobuf/
# tensor_bundle.proto (BundleHeaderProto on the empty key, BundleEntryProto per tensor) and tensorflow/core/util/
# tensor_bundle/naming.cc (<prefix>.data-%05d-of-%05d).  Variable names are the ones TF 1.12 gives the reference's graph:
# utils/tf_util.py:473-479 creates beta / gamma with tf.Variable inside variable_scope('<layer>/bn') and lets
# tf.train.ExponentialMovingAverage shadow the OUTPUT TENSORS of tf.nn.moments; slot_creator opens
# variable_scope(None, primary.op.name + '/ExponentialMovingAverage') INSIDE that scope, so the shadow of
# '<layer>/bn/moments/Squeeze' is the variable '<layer>/bn/<layer>/bn/moments/Squeeze/ExponentialMovingAverage' (the scope
# appears twice); AdamOptimizer (train_n_est_w_experts.py:182) adds '<var>/Adam', '<var>/Adam_1', 'beta1_power', 'beta2_power'.
def _vint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _pb_entry(dtype, shape, shard, offset, size, crc_masked):
    dims = b"".join(b"\x12" + _vint(len(d)) + d for d in (b"\x08" + _vint(n) for n in shape))     # repeated Dim {size = 1}
    msg = b"\x08" + _vint(dtype) + b"\x12" + _vint(len(dims)) + dims
    if shard:
        msg += b"\x18" + _vint(shard)
    if offset:
        msg += b"\x20" + _vint(offset)
    return msg + b"\x28" + _vint(size) + b"\x35" + struct.pack("<I", crc_masked)


class _HandTable:
    """A LevelDB table built straight from the format description: small blocks, restart points every 3 entries."""

    def __init__(self, crc, mask, block_bytes=220, restart_every=3):
        self.crc, self.mask, self.block_bytes, self.every = crc, mask, block_bytes, restart_every
        self.file = bytearray()
        self.index = []                        # (last key of the block, offset, size)
        self._reset()

    def _reset(self):
        self.buf, self.restarts, self.n, self.last = bytearray(), [0], 0, b""

    def add(self, key, value):
        shared = 0
        if self.n and self.n % self.every == 0:
            self.restarts.append(len(self.buf))              # a restart point stores the full key
        elif self.n:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _vint(shared) + _vint(len(key) - shared) + _vint(len(value)) + key[shared:] + value
        self.last, self.n = key, self.n + 1
        if len(self.buf) >= self.block_bytes:
            self.flush()

    def _emit(self, block):
        off = len(self.file)
        self.file += block + b"\x00" + struct.pack("<I", self.mask(self.crc(bytes(block) + b"\x00")))
        return off, len(block)

    def flush(self):
        if not self.n:
            return
        block = bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))
        self.index.append((self.last,) + self._emit(block))
        self._reset()

    def finish(self):
        self.flush()
        meta = self._emit(struct.pack("<II", 0, 1))          # empty metaindex block: one restart, no entries
        ib = bytearray()
        rs = []
        for key, off, size in self.index:                    # index block: restart interval 1, value = BlockHandle
            rs.append(len(ib))
            h = _vint(off) + _vint(size)
            ib += b"\x00" + _vint(len(key)) + _vint(len(h)) + key + h
        idx = self._emit(bytes(ib) + b"".join(struct.pack("<I", r) for r in rs) + struct.pack("<I", len(rs)))
        footer = (_vint(meta[0]) + _vint(meta[1]) + _vint(idx[0]) + _vint(idx[1])).ljust(40, b"\x00")
        return bytes(self.file) + footer + struct.pack("<Q", 0xdb4775248b80fb57)


def _tf112_names(layer, bn=True):
    names = [layer + "/weights", layer + "/biases"]
    if bn:
        names += [layer + "/bn/beta", layer + "/bn/gamma",
                  "%s/bn/%s/bn/moments/Squeeze/ExponentialMovingAverage" % (layer, layer),
                  "%s/bn/%s/bn/moments/Squeeze_1/ExponentialMovingAverage" % (layer, layer)]
    return names


def _hand_checkpoint(tmp_path, layers, extra=()):
    """-> (prefix, {tf name: array}).  Trainable variables get Adam slots; tensors alternate between two data shards."""
    from nesti_net_amd import tf_ckpt
    rng = np.random.RandomState(5)
    tensors = {}
    for layer, (wshape, cout, bn) in layers.items():
        for nm in _tf112_names(layer, bn):
            tensors[nm] = rng.randn(*(wshape if nm.endswith("/weights") else (cout,))).astype(np.float32)
            if "/moments/" not in nm:
                tensors[nm + "/Adam"] = rng.randn(*tensors[nm].shape).astype(np.float32)
                tensors[nm + "/Adam_1"] = np.abs(rng.randn(*tensors[nm].shape)).astype(np.float32)
    tensors["beta1_power"] = np.float32(0.9 ** 1000).reshape(())
    tensors["beta2_power"] = np.float32(0.999 ** 1000).reshape(())
    tensors["Variable"] = np.array(123456, dtype=np.int32).reshape(())          # the `batch` step counter, train_n_est_w_experts.py:128
    for k, v in extra:
        tensors[k] = v
    prefix = str(tmp_path / "model.ckpt")
    shards = [bytearray(), bytearray()]
    tb = _HandTable(tf_ckpt.crc32c_py, tf_ckpt.mask_crc)
    tb.add(b"", b"\x08\x02\x1a\x02\x08\x01")                   # BundleHeaderProto: num_shards = 2, version { producer = 1 }
    for i, name in enumerate(sorted(tensors)):                # a table's keys are sorted bytewise
        a = tensors[name]
        sid = i % 2
        raw = a.tobytes()
        off = len(shards[sid])
        shards[sid] += raw
        dt = {np.dtype(np.float32): 1, np.dtype(np.int32): 3}[a.dtype]
        tb.add(name.encode(), _pb_entry(dt, a.shape, sid, off, len(raw), tf_ckpt.mask_crc(tf_ckpt.crc32c_py(raw))))
    open(prefix + ".index", "wb").write(tb.finish())
    for sid in range(2):
        open("%s.data-%05d-of-00002" % (prefix, sid), "wb").write(bytes(shards[sid]))
    assert len(tb.index) >= 5, "the index must span several data blocks"
    return prefix, tensors


def test_tf112_shaped_checkpoint_assembled_by_hand(tmp_path):
    """Several data blocks, restart points, prefix-compressed keys, Adam slots, the doubled EMA shadow names, two data
    shards, multi-byte varints: the reader maps every graph variable -- and refuses the corrupted variants loudly."""
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import tf_ckpt
    layers = {"inception1Expert_6_conv1": ((1, 1, 1, 60, 42), 42, True), "inception1Expert_6_conv2": ((3, 3, 3, 42, 21), 21, True),
              "fc3Expert_6": ((128, 64), 64, True), "fc4Expert_6": ((64, 3), 3, False)}
    prefix, tensors = _hand_checkpoint(tmp_path, layers)
    ents = tf_ckpt.read_index(prefix + ".index")
    assert sorted(ents) == sorted(tensors) and {e["shard_id"] for e in ents.values()} == {0, 1}
    assert max(e["offset"] for e in ents.values()) > 16384                       # three-byte varints in the entry protos
    raw = tf_ckpt.read_bundle(prefix)
    for k, v in tensors.items():
        assert np.array_equal(raw[k], v) and raw[k].dtype == v.dtype, k
    expected = {}
    for layer, (wshape, cout, bn) in layers.items():
        expected[layer + "/weights"], expected[layer + "/biases"] = wshape, (cout,)
        if bn:
            for s in ("beta", "gamma", "mean", "var"):
                expected["%s/bn/%s" % (layer, s)] = (cout,)
    got = tf_ckpt.map_variables(raw, expected)
    assert sorted(got) == sorted(expected)
    for layer, (_, _, bn) in layers.items():
        assert np.array_equal(got[layer + "/weights"], tensors[layer + "/weights"])
        if bn:
            sq = "%s/bn/%s/bn/moments/Squeeze" % (layer, layer)
            assert np.array_equal(got[layer + "/bn/mean"], tensors[sq + "/ExponentialMovingAverage"])
            assert np.array_equal(got[layer + "/bn/var"], tensors[sq + "_1/ExponentialMovingAverage"])
            assert not np.array_equal(got[layer + "/bn/beta"], tensors[layer + "/bn/beta/Adam"])      # slots are not variables
    # ---- refusals -----------------------------------------------------------------------------------------------------
    d = str(tmp_path / "bad")
    os.makedirs(d)
    # (a) a third shadow under one bn scope (e.g. a second tower sharing the scope): mean / variance are ambiguous
    p2, _ = _hand_checkpoint(tmp_path / "bad", layers, extra=[
        ("fc3Expert_6/bn/fc3Expert_6/bn/moments/Squeeze_2/ExponentialMovingAverage", np.zeros(64, np.float32))])
    with pytest.raises(KeyError, match="expected 2 EMA shadow"):
        tf_ckpt.map_variables(tf_ckpt.read_bundle(p2), expected)
    # (b) a shard is missing
    os.remove(p2 + ".data-00001-of-00002")
    with pytest.raises((FileNotFoundError, ValueError)):
        tf_ckpt.read_bundle(p2)
    # (c) one flipped bit inside the SECOND shard
    s1 = bytearray(open(prefix + ".data-00001-of-00002", "rb").read())
    s1[len(s1) // 2] ^= 0x10
    open(prefix + ".data-00001-of-00002", "wb").write(bytes(s1))
    with pytest.raises(ValueError, match="fails its crc32c"):
        tf_ckpt.read_bundle(prefix)
    # (d) one flipped bit inside a data block in the MIDDLE of the index
    ix = bytearray(open(prefix + ".index", "rb").read())
    ix[len(ix) // 3] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(ix))
    with pytest.raises(ValueError, match="crc32c"):
        tf_ckpt.read_index(prefix + ".index")


def test_header_shard_count_is_honoured(tmp_path):
    """Every tensor the graph needs may sit in shard 0 while the bundle still has two shards (the header says so): the data
    file names carry the HEADER's shard count, not 1 + the largest shard id among the entries."""
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import tf_ckpt
    tb = _HandTable(tf_ckpt.crc32c_py, tf_ckpt.mask_crc)
    a = np.arange(6, dtype=np.float32)
    tb.add(b"", b"\x08\x02\x1a\x02\x08\x01")                   # num_shards = 2
    tb.add(b"fc/biases", _pb_entry(1, (6,), 0, 0, 24, tf_ckpt.mask_crc(tf_ckpt.crc32c_py(a.tobytes()))))
    prefix = str(tmp_path / "m.ckpt")
    open(prefix + ".index", "wb").write(tb.finish())
    open(prefix + ".data-00000-of-00002", "wb").write(a.tobytes())
    open(prefix + ".data-00001-of-00002", "wb").write(b"")
    assert np.array_equal(tf_ckpt.read_bundle(prefix)["fc/biases"], a)
