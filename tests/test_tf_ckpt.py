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
