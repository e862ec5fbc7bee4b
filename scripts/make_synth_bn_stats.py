"""Batch-norm statistics for the synthetic weights, matched to the data the way training would leave them.

``weights.synthetic_weights`` draws He-scaled random weights; with RANDOM batch-norm EMA statistics the
activations of a 20-layer ReLU stack are dominated by their (input-independent) mean, the gate's logits barely
depend on the query and ``calibrate_gate`` has to amplify tiny differences -- which amplifies the rounding noise of a
16-bit run by the same factor, so arg-max parity between dtypes measured on such weights says little about a
trained network.  A trained Nesti-Net carries the EMA of the batch statistics (``utils/tf_util.py:478-494``): every
layer's pre-activation is zero-mean / unit-variance over the data.  This script reproduces that state: one forward
pass of the CPU oracle over a sample of real MuPS tensors with batch statistics (training-mode BN,
``utils/tf_util.py:488-490``), recording each layer's per-channel mean / variance as its EMA.

Runs in the build container (CPU, about a minute); the result is data (per-channel float32 statistics), committed as
``nesti-net_amd/data/synth_bn_stats.npz`` and picked up by ``weights.synthetic_weights`` for the default
configuration.  Test / bench infrastructure only: it imports ``oracle/``.

    python scripts/make_synth_bn_stats.py
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import synth, weights  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from oracle import mups_ref, net_ref, patches_ref  # noqa: E402

SAMPLE = [  # (cloud kwargs, number of queries strided over the cloud)
    (dict(shape="ellipsoid", n=100000, seed=1234), 96),
    (dict(shape="torus", n=40000, seed=1237, noise=0.00125, density="gradient"), 32),
    (dict(shape="sphere", n=40000, seed=1238, noise=0.012, density="striped"), 32),
    (dict(shape="box", n=20000, seed=1236, noise=0.006), 32),
]


def sample_mups(cfg):
    out = []
    for kw, m in SAMPLE:
        pts, _ = synth.make_cloud(**kw)
        q = np.arange(len(pts) // (2 * m), len(pts), len(pts) // m)[:m]
        _, r_abs = patches_ref.patch_radii(pts, cfg.patch_radius)
        p, n, _, _ = patches_ref.extract_patches(pts, q, r_abs, cfg.num_point, 3627473)
        out.append(mups_ref.mups_assemble(p, n, cfg.n_scales, dtype=np.float32))
    return np.concatenate(out)


def main():
    cfg = NestiConfig()
    W = weights.synthetic_weights(cfg, bn="random")
    mups = torch.as_tensor(sample_mups(cfg))
    print("sample MuPS", tuple(mups.shape))
    stats = {}
    orig_bn = net_ref.batch_norm

    def bn_train(x, Wd, scope, dtype):
        red = tuple(range(x.dim() - 1))
        mean = x.mean(dim=red)
        var = x.var(dim=red, unbiased=False)                     # tf.nn.moments (utils/tf_util.py:477)
        Wd[scope + "/bn/mean"] = mean.to(torch.float32).numpy()
        Wd[scope + "/bn/var"] = np.maximum(var.to(torch.float32).numpy(), 1e-6)
        stats[scope + "/bn/mean"], stats[scope + "/bn/var"] = Wd[scope + "/bn/mean"], Wd[scope + "/bn/var"]
        return orig_bn(x, Wd, scope, dtype)

    net_ref.batch_norm = bn_train
    torch.set_num_threads(os.cpu_count() or 1)
    with torch.no_grad():
        probs, logits = net_ref.gate_forward(mups, W, torch.float32)
        print("gate logits: mean over queries", logits.mean(0).numpy().round(3), "std over queries", logits.std(0).numpy().round(3))
        for e in range(cfg.n_experts):
            lo = min(cfg.expert_dict[e]) * 20
            hi = lo + 20 * len(cfg.expert_dict[e])
            n = net_ref.expert_forward(mups[..., lo:hi], W, e, torch.float32)
            print("expert", e, "output std over queries", n.std(0).numpy().round(3))
    net_ref.batch_norm = orig_bn
    out = os.path.join(REPO, "nesti-net_amd", "data", "synth_bn_stats.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    np.savez_compressed(out, seed=weights.WEIGHT_SEED, **{k: v.astype(np.float32) for k, v in stats.items()})
    print(out, "%d tensors, %.1f KB" % (len(stats), os.path.getsize(out) / 1024))


if __name__ == "__main__":
    main()
