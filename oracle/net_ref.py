"""ORACLE (test infrastructure, NOT product code): CPU restatement of the
Nesti-Net mixture-of-experts graph on torch-CPU tensors.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.

Parity status: **unpinned at the TensorFlow boundary** (TF 1.12 absent; the
conv / pool / BN / softmax kernels live in that un-vendored dependency, call
sites ``utils/tf_util.py:298,340,424,450,494``, ``models/experts_n_est.py:177``).
The TF semantics restated here (SAME padding for even kernels, valid-count
average pooling, BN eps 1e-3) are pinned by the known-answer tests in
``tests/test_oracle_net.py`` against naive loop implementations.

Weights are a dict name -> float32 ndarray in TF variable layout:
  conv  ``<scope>/weights`` [kd,kh,kw,Cin,Cout], ``<scope>/biases`` [Cout]
        (``utils/tf_util.py:289-302``), ``<scope>/bn/{beta,gamma,mean,var}``
        (``utils/tf_util.py:473-479``; mean/var are the EMA shadows)
  fc    ``<scope>/weights`` [In,Out], ``<scope>/biases`` (``utils/tf_util.py:332-342``)
"""
import numpy as np
import torch
import torch.nn.functional as F

TRAINED_EXPERT_DICT = {0: [0], 1: [0], 2: [1], 3: [1], 4: [2], 5: [2], 6: [0, 1, 2]}  # train_n_est_w_experts.py:62
BN_EPS = 1e-3  # utils/tf_util.py:494


def _same_pad(k):
    """TF SAME at stride 1: total k-1, floor on the low side."""
    lo = (k - 1) // 2
    return lo, (k - 1) - lo


def _t(a, dtype):
    return torch.as_tensor(np.asarray(a)).to(dtype)


def conv3d_same_direct(x, w, b):
    """x [B,D,H,W,C] channels-last, w [k,k,k,Cin,Cout] (``utils/tf_util.py:298-302``), through ``F.conv3d``."""
    k = w.shape[0]
    lo, hi = _same_pad(k)
    xc = x.permute(0, 4, 1, 2, 3)
    xc = F.pad(xc, (lo, hi, lo, hi, lo, hi))
    y = F.conv3d(xc, w.permute(4, 3, 0, 1, 2), b)
    return y.permute(0, 2, 3, 4, 1)


def conv3d_same_taps(x, w, b):
    """The same convolution as a sum over the k^3 taps of [positions, Cin] @ [Cin, Cout] on the zero-padded volume --
    the definition itself, and 6-12x faster than torch's fp64 conv3d (which unfolds the input 125-fold for a 5^3 kernel);
    equal to :func:`conv3d_same_direct` to rounding (``tests/test_oracle_net.py``)."""
    k = w.shape[0]
    lo, hi = _same_pad(k)
    B, D, H, Wd, C = x.shape
    xp = F.pad(x, (0, 0, lo, hi, lo, hi, lo, hi))
    out = torch.zeros((B * D * H * Wd, w.shape[4]), dtype=x.dtype)
    for a in range(k):
        for bb in range(k):
            for c in range(k):
                out.addmm_(xp[:, a:a + D, bb:bb + H, c:c + Wd, :].reshape(-1, C), w[a, bb, c])
    return (out + b).reshape(B, D, H, Wd, -1)


def conv3d_same(x, w, b):
    """fp64 (the parity side of the GPU tests): tap sum; fp32 (bench.py's cpu_baseline leg, which should run what a CPU
    user of the reference would run): the library convolution."""
    if x.dtype == torch.float64 and w.shape[0] > 1:
        return conv3d_same_taps(x, w, b)
    return conv3d_same_direct(x, w, b)


def avg_pool3d_same(x, k):
    """``tf.nn.avg_pool3d`` k^3, stride 1, SAME: mean over the taps that fall
    inside the volume (``utils/tf_util.py:450-454``)."""
    if k == 1:
        return x
    lo, hi = _same_pad(k)
    xc = F.pad(x.permute(0, 4, 1, 2, 3), (lo, hi, lo, hi, lo, hi))
    s = F.avg_pool3d(xc, k, stride=1) * float(k ** 3)
    ones = F.pad(torch.ones((1, 1) + tuple(x.shape[1:4]), dtype=x.dtype), (lo, hi, lo, hi, lo, hi))
    cnt = F.avg_pool3d(ones, k, stride=1) * float(k ** 3)
    return (s / cnt).permute(0, 2, 3, 4, 1)


def max_pool3d_2(x):
    """``tf.nn.max_pool3d`` 2^3 stride 2 SAME on even sizes (``utils/tf_util.py:424-428``)."""
    return F.max_pool3d(x.permute(0, 4, 1, 2, 3), 2, 2).permute(0, 2, 3, 4, 1)


def batch_norm(x, W, scope, dtype):
    """Inference branch of ``utils/tf_util.py:491-494``."""
    mean, var = _t(W[scope + "/bn/mean"], dtype), _t(W[scope + "/bn/var"], dtype)
    beta, gamma = _t(W[scope + "/bn/beta"], dtype), _t(W[scope + "/bn/gamma"], dtype)
    return (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta


def conv_bn_relu(x, W, scope, dtype):
    y = conv3d_same(x, _t(W[scope + "/weights"], dtype), _t(W[scope + "/biases"], dtype))
    return torch.relu(batch_norm(y, W, scope, dtype))           # utils/tf_util.py:304-310


def fc(x, W, scope, dtype, bn=True, relu=True):
    y = x @ _t(W[scope + "/weights"], dtype) + _t(W[scope + "/biases"], dtype)   # utils/tf_util.py:340-343
    if bn:
        y = batch_norm(y, W, scope, dtype)
    return torch.relu(y) if relu else y


def inception(x, W, scope, k0, k1, dtype):
    """``models/experts_n_est.py:294-314``."""
    c1 = conv_bn_relu(x, W, scope + "_conv1", dtype)
    c2 = conv_bn_relu(c1, W, scope + "_conv2", dtype)
    c3 = conv_bn_relu(c1, W, scope + "_conv3", dtype)
    c4 = conv_bn_relu(avg_pool3d_same(x, k0), W, scope + "_conv4", dtype)
    assert W[scope + "_conv2/weights"].shape[0] == k0 and W[scope + "_conv3/weights"].shape[0] == k1
    return torch.cat([c1, c2, c3, c4], dim=4)


def max_pool3d_3s2_same(x):
    """``tf.nn.max_pool3d`` [3,3,3] stride 2 SAME (``models/experts_n_est.py:238``): out = ceil(in/2); for in = 3
    TF pads one voxel on each side (ignored by the max), i.e. torch's padding=1 with -inf."""
    assert x.shape[1] == x.shape[2] == x.shape[3] == 3
    return F.max_pool3d(x.permute(0, 4, 1, 2, 3), 3, 2, padding=1).permute(0, 2, 3, 4, 1)


def conv_net_3g(x, W, s, dtype):
    """``conv_net_3g`` (``models/experts_n_est.py:217-240``): [B,3,3,3,C] -> [B, 2*2*2*1536]."""
    x = inception(x, W, "inception1" + s, 2, 3, dtype)
    x = inception(x, W, "inception2" + s, 2, 3, dtype)
    x = inception(x, W, "inception3" + s, 1, 2, dtype)
    x = inception(x, W, "inception4" + s, 1, 2, dtype)
    x = max_pool3d_3s2_same(x)
    return x.reshape(x.shape[0], -1)


def gate_forward(mups, W, dtype=torch.float64):
    """``scale_manager_net`` + ``conv_net_8g`` / ``conv_net_3g`` (``models/experts_n_est.py:155-240``).
    Returns (probs [B,E], logits-after-relu [B,E])."""
    s = "gating_conv"
    x = _t(mups, dtype)
    if x.shape[1] == 3:                                                # 27 Gaussians  :162-163
        g = conv_net_3g(x, W, s, dtype)
        g = fc(g, W, "fc1noise", dtype)
        g = fc(g, W, "fc2noise", dtype)
        g = fc(g, W, "fc3noise", dtype)
        logits = fc(g, W, "fc4noise", dtype, bn=False, relu=True)
        return torch.softmax(logits, dim=1), logits
    x = inception(x, W, "inception1" + s, 3, 5, dtype)
    x = inception(x, W, "inception2" + s, 3, 5, dtype)
    x = inception(x, W, "inception3" + s, 3, 5, dtype)
    x = max_pool3d_2(x)
    x = inception(x, W, "inception5" + s, 2, 4, dtype)
    x = inception(x, W, "inception6" + s, 2, 4, dtype)
    x = max_pool3d_2(x)
    x = inception(x, W, "inception8" + s, 1, 2, dtype)
    x = max_pool3d_2(x)
    g = x.reshape(x.shape[0], -1)
    g = fc(g, W, "fc1noise", dtype)
    g = fc(g, W, "fc2noise", dtype)
    g = fc(g, W, "fc3noise", dtype)
    logits = fc(g, W, "fc4noise", dtype, bn=False, relu=True)     # :174  relu BEFORE softmax
    return torch.softmax(logits, dim=1), logits                    # :177


def expert_forward(mups_slice, W, i, dtype=torch.float64):
    """``normal_est_net`` 8^3 branch (``models/experts_n_est.py:243-291``)."""
    s = "Expert_%d" % i
    x = _t(mups_slice, dtype)
    if x.shape[1] == 3:                                                # :275-276 (`divider` unused on this branch)
        g = conv_net_3g(x, W, s + "_expert_conv", dtype)
        g = fc(g, W, "fc1" + s, dtype)
        g = fc(g, W, "fc2" + s, dtype)
        g = fc(g, W, "fc3" + s, dtype)
        return fc(g, W, "fc4" + s, dtype, bn=False, relu=False)
    x = inception(x, W, "inception1" + s, 3, 5, dtype)
    x = inception(x, W, "inception2" + s, 3, 5, dtype)
    x = max_pool3d_2(x)
    x = inception(x, W, "inception4" + s, 2, 4, dtype)
    x = max_pool3d_2(x)
    x = inception(x, W, "inception6" + s, 2, 4, dtype)
    x = max_pool3d_2(x)
    g = x.reshape(x.shape[0], -1)
    g = fc(g, W, "fc1" + s, dtype)
    g = fc(g, W, "fc2" + s, dtype)
    g = fc(g, W, "fc3" + s, dtype)
    return fc(g, W, "fc4" + s, dtype, bn=False, relu=False)       # :286


def single_forward(mups, W, dtype=torch.float64):
    """``ss_norm_est.get_model`` after the 3DmFV (``models/ss_norm_est.py:49-92``): one scale
    [B,8,8,8,20] -> normal [B,3].  Dropout (``:76-85``) is the identity at inference."""
    x = _t(mups, dtype)
    x = inception(x, W, "inception1", 3, 5, dtype)
    x = inception(x, W, "inception2", 3, 5, dtype)
    x = inception(x, W, "inception3", 3, 5, dtype)
    x = max_pool3d_2(x)
    x = inception(x, W, "inception5", 3, 5, dtype)
    x = inception(x, W, "inception6", 3, 5, dtype)
    x = max_pool3d_2(x)
    g = x.reshape(x.shape[0], -1)                                  # :66  [B, 2*2*2*1536], voxel-major
    g = fc(g, W, "fc1", dtype)
    g = fc(g, W, "fc2", dtype)
    g = fc(g, W, "fc3", dtype)
    return fc(g, W, "fc4", dtype, bn=False, relu=False)            # :86


def _ss_tower(x, W, sfx, dtype, last_relu):
    """The tower shared by ``noise_est_net`` / ``normal_est_net`` of ``models/ms_sw_n_est.py:139-215``
    (= ``single_forward`` with scope suffix ``sfx``)."""
    x = inception(x, W, "inception1" + sfx, 3, 5, dtype)
    x = inception(x, W, "inception2" + sfx, 3, 5, dtype)
    x = inception(x, W, "inception3" + sfx, 3, 5, dtype)
    x = max_pool3d_2(x)
    x = inception(x, W, "inception5" + sfx, 3, 5, dtype)
    x = inception(x, W, "inception6" + sfx, 3, 5, dtype)
    x = max_pool3d_2(x)
    g = x.reshape(x.shape[0], -1)
    g = fc(g, W, "fc1" + sfx, dtype)
    g = fc(g, W, "fc2" + sfx, dtype)
    g = fc(g, W, "fc3" + sfx, dtype)
    return fc(g, W, "fc4" + sfx, dtype, bn=False, relu=last_relu)


SWITCH_THRESHOLD = 0.015   # models/ms_sw_n_est.py:80


def switch_forward(mups, W, dtype=torch.float64):
    """``ms_sw_n_est.get_model`` after the two 3DmFVs (``models/ms_sw_n_est.py:75-82``): MuPS
    [B,8,8,8,40] (small scale = channels 0..19, large = 20..39) ->
    dict(noise [B], pick [B] (0 small / 1 large), normals [B,3], n_small, n_large)."""
    x = _t(mups, dtype)
    small, large = x[..., 0:20], x[..., 20:40]
    noise = _ss_tower(large, W, "noise", dtype, last_relu=True)[:, 0]      # :75, relu on fc4 :172
    n_large = _ss_tower(large, W, "large", dtype, last_relu=False)         # :77
    n_small = _ss_tower(small, W, "small", dtype, last_relu=False)         # :78
    mask = noise < SWITCH_THRESHOLD                                        # :80
    normals = torch.where(mask[:, None], n_small, n_large)                 # :82
    return {"noise": noise, "pick": (~mask).to(torch.int64), "normals": normals, "n_small": n_small, "n_large": n_large}


def multi_forward(mups, W, n_scales, dtype=torch.float64):
    """``ms_norm_est.get_model`` after the 3DmFV (``models/ms_norm_est.py:74-140``): MuPS
    [B,8,8,8,20*S] -> normal [B,3].  Scopes carry the last scale index (``'inception_s'+str(s)``, ``:79``)."""
    sc = "inception_s%d_l_" % (n_scales - 1)
    x = _t(mups, dtype)
    x = inception(x, W, sc + "1", 3, 5, dtype)
    x = inception(x, W, sc + "2", 3, 5, dtype)
    x = inception(x, W, sc + "3", 3, 5, dtype)
    x = max_pool3d_2(x)
    x = inception(x, W, sc + "5", 3, 4, dtype)                     # :87-89  kernel_sizes=[3, 4]
    x = inception(x, W, sc + "6", 3, 4, dtype)
    x = max_pool3d_2(x)
    g = x.reshape(x.shape[0], -1)
    g = fc(g, W, "fc1", dtype)
    g = fc(g, W, "fc2", dtype)
    g = fc(g, W, "fc3", dtype)
    return fc(g, W, "fc4", dtype, bn=False, relu=False)


def moe_forward(mups, W, expert_dict=None, dtype=torch.float64, top1_only=False):
    """``get_model`` after MuPS (``models/experts_n_est.py:78-108``) plus the
    driver's arg-max / select (``test_n_est_w_experts.py:150-152``).

    Returns dict(probs [B,E], expert [B] int64, normals [B,3], n_est [E,B,3] or None).
    ``top1_only`` evaluates only the selected expert per point (output-identical)."""
    expert_dict = expert_dict or TRAINED_EXPERT_DICT
    E = len(expert_dict)
    mups_t = _t(mups, dtype)
    B = mups_t.shape[0]
    probs, logits = gate_forward(mups_t, W, dtype)
    expert = torch.argmax(probs, dim=1)      # np.argmax(axis=0) on [E,B]: first index on ties
    # torch.argmax does not promise first-index on ties; enforce it.
    pmax = probs.max(dim=1, keepdim=True).values
    expert = torch.argmax((probs == pmax).to(torch.int8), dim=1)
    n_est = None
    normals = torch.zeros(B, 3, dtype=dtype)

    def run(i, rows):
        lo = min(expert_dict[i]) * 20                               # experts_n_est.py:100-101
        hi = lo + 20 * len(expert_dict[i])
        return expert_forward(mups_t[rows][..., lo:hi], W, i, dtype)

    if top1_only:
        for i in range(E):
            rows = torch.nonzero(expert == i).flatten()
            if len(rows):
                normals[rows] = run(i, rows)
    else:
        allrows = torch.arange(B)
        n_est = torch.stack([run(i, allrows) for i in range(E)])    # :105
        normals = n_est[expert, allrows]                            # test_n_est_w_experts.py:152
    return {"probs": probs, "logits": logits, "expert": expert, "normals": normals, "n_est": n_est}


# ---------------------------------------------------------------------------
# naive loop implementations used ONLY to pin the semantics above (tiny sizes)
# ---------------------------------------------------------------------------
def over_chunks(fn, n_rows, chunk=16, workers=None, threads=4):
    """``[fn(slice) for consecutive slices of `chunk` rows]`` with the slices spread over a few Python threads, each torch call on
    ``threads`` intra-op threads.  Test plumbing, not arithmetic: torch's fp64 conv3d on the CPU gets SLOWER beyond ~16 threads
    (measured on a 256-core host: 128 queries in 19 s on 16 threads, 25 s on 32, 37 s on 64), while independent chunks scale; the GIL is
    released inside the operators.  Results may differ from a single-chunk run in the last bits of an fp64 (BLAS blocking), nothing a
    tolerance of these tests can see."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    spans = [slice(i, min(n_rows, i + chunk)) for i in range(0, n_rows, chunk)]
    if workers is None:
        workers = max(1, min(16, (os.cpu_count() or 1) // threads))
    workers = min(workers, len(spans))
    if workers <= 1:
        return [fn(sp) for sp in spans]
    before = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        with ThreadPoolExecutor(workers) as ex:
            return list(ex.map(fn, spans))
    finally:
        torch.set_num_threads(before)


def naive_conv3d_same(x, w, b):
    x, w = np.asarray(x, np.float64), np.asarray(w, np.float64)
    B, D, H, Wd, C = x.shape
    k = w.shape[0]
    lo, _ = _same_pad(k)
    y = np.zeros((B, D, H, Wd, w.shape[4]))
    for z in range(D):
        for yy in range(H):
            for xx in range(Wd):
                for a in range(k):
                    for bb in range(k):
                        for c in range(k):
                            iz, iy, ix = z + a - lo, yy + bb - lo, xx + c - lo
                            if 0 <= iz < D and 0 <= iy < H and 0 <= ix < Wd:
                                y[:, z, yy, xx] += x[:, iz, iy, ix] @ w[a, bb, c]
    return y + np.asarray(b, np.float64)


def naive_avg_pool3d_same(x, k):
    x = np.asarray(x, np.float64)
    B, D, H, Wd, C = x.shape
    lo, _ = _same_pad(k)
    y = np.zeros_like(x)
    for z in range(D):
        for yy in range(H):
            for xx in range(Wd):
                acc, cnt = 0.0, 0
                for a in range(k):
                    for bb in range(k):
                        for c in range(k):
                            iz, iy, ix = z + a - lo, yy + bb - lo, xx + c - lo
                            if 0 <= iz < D and 0 <= iy < H and 0 <= ix < Wd:
                                acc = acc + x[:, iz, iy, ix]
                                cnt += 1
                y[:, z, yy, xx] = acc / cnt
    return y
