"""Pins the TF layer semantics restated in oracle/net_ref.py (CPU only): SAME padding for
even kernels, valid-count average pooling, BN inference, first-index arg-max."""
import numpy as np
import torch

from oracle import net_ref as N


def test_same_padding_even_and_odd_kernels():
    rng = np.random.RandomState(0)
    for k in (1, 2, 3, 4, 5):
        x = rng.randn(2, 4, 4, 4, 3)
        w = rng.randn(k, k, k, 3, 5)
        b = rng.randn(5)
        ref = N.naive_conv3d_same(x, w, b)
        # the fp64 dispatcher (tap sum for k > 1), both implementations, and the fp32 dispatcher (library convolution)
        for fn in (N.conv3d_same, N.conv3d_same_taps, N.conv3d_same_direct):
            got = fn(torch.as_tensor(x), torch.as_tensor(w), torch.as_tensor(b)).numpy()
            assert np.abs(got - ref).max() < 1e-10, (k, fn.__name__)
        got32 = N.conv3d_same(torch.as_tensor(x).float(), torch.as_tensor(w).float(), torch.as_tensor(b).float()).numpy()
        assert np.abs(got32 - ref).max() < 1e-4, k


def test_same_padding_k2_is_high_side():
    """k=2 SAME: pad 0 before, 1 after -> out[v] = w0*x[v] + w1*x[v+1]."""
    x = np.zeros((1, 2, 2, 2, 1))
    x[0, 1, 1, 1, 0] = 1.0
    w = np.zeros((2, 2, 2, 1, 1))
    w[1, 1, 1, 0, 0] = 7.0
    y = N.conv3d_same(torch.as_tensor(x), torch.as_tensor(w), torch.zeros(1, dtype=torch.float64)).numpy()
    assert y[0, 0, 0, 0, 0] == 7.0 and np.count_nonzero(y) == 1


def test_avg_pool_divides_by_valid_count():
    rng = np.random.RandomState(1)
    for k in (2, 3):
        x = rng.randn(1, 4, 4, 4, 2)
        got = N.avg_pool3d_same(torch.as_tensor(x), k).numpy()
        assert np.abs(got - N.naive_avg_pool3d_same(x, k)).max() < 1e-12
    ones = torch.ones(1, 8, 8, 8, 1, dtype=torch.float64)
    assert torch.allclose(N.avg_pool3d_same(ones, 3), ones)     # padding excluded from the divisor


def test_max_pool():
    x = torch.arange(64, dtype=torch.float64).reshape(1, 4, 4, 4, 1)
    y = N.max_pool3d_2(x)
    assert y.shape == (1, 2, 2, 2, 1) and y[0, 0, 0, 0, 0] == 21 and y[0, 1, 1, 1, 0] == 63


def test_max_pool_3_stride2_same_on_3cubed():
    """tf.nn.max_pool3d([3,3,3], stride 2, SAME) on 3^3 (models/experts_n_est.py:238): TF's SAME rule gives
    out = ceil(3/2) = 2 and pad_total = (2-1)*2 + 3 - 3 = 2 -> one voxel in front, so cell o covers {o, o+1}."""
    x = torch.arange(2 * 27 * 3, dtype=torch.float64).reshape(2, 3, 3, 3, 3)
    x = torch.sin(x)
    y = N.max_pool3d_3s2_same(x).numpy()
    assert y.shape == (2, 2, 2, 2, 3)
    xn = x.numpy()
    for z in range(2):
        for yy in range(2):
            for xx in range(2):
                ref = xn[:, z:z + 2, yy:yy + 2, xx:xx + 2].max(axis=(1, 2, 3))
                assert np.array_equal(y[:, z, yy, xx], ref)


def test_bn_inference_eps():
    W = {"s/bn/mean": np.array([1.0]), "s/bn/var": np.array([0.0]), "s/bn/beta": np.array([0.5]),
         "s/bn/gamma": np.array([2.0])}
    y = N.batch_norm(torch.tensor([[2.0]], dtype=torch.float64), W, "s", torch.float64)
    assert abs(float(y) - (1.0 / np.sqrt(1e-3) * 2.0 + 0.5)) < 1e-9     # eps = 1e-3 (utils/tf_util.py:494)
