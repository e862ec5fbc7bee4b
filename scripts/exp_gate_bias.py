"""Is part of the plain-f16 gate's error on a logit difference SYSTEMATIC (a per-expert constant offset: weight-rounding errors
times the non-zero mean of post-ReLU activations), and what would removing it buy?  f16 gate vs f16x3 gate on 20k strided queries
of the bench cloud (calibrated routing); logit differences from log-probabilities (softmax is shift-invariant).
-> gpurun_out/exp_gate_bias.txt"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import synth, weights  # noqa: E402
from nesti_net_amd.calibrate import calibrate_gate  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.model import NestiNet  # noqa: E402
from nesti_net_amd.provider import CloudPatches  # noqa: E402

dev = torch.device("cuda:0")
cfg = NestiConfig()
out = []
for name, kw in (("ellipsoid", dict(n=100000, seed=1234)), ("torus_noise", dict(shape="torus", n=100000, seed=78, noise=0.00125)),
                 ("box_striped", dict(shape="box", n=100000, seed=83, density="striped"))):
    pts = synth.make_cloud(**({"shape": "ellipsoid"} | kw))[0]
    if name == "ellipsoid":
        cp = CloudPatches(pts, cfg, device=dev)
        sp, sn = cp.build(0, 512)
        W = calibrate_gate(cfg, weights.synthetic_weights(cfg), sp, sn, device=dev)
        del cp, sp, sn
    Q = 20000
    q = np.arange(3, len(pts), len(pts) // Q)[:Q]
    cp = CloudPatches(pts, cfg, device=dev, pidx=q)
    logp = {}
    for mode in ("f16", "f16x3"):
        net = NestiNet(cfg, W, dtype=mode, device=dev, max_batch=4096)
        ps = []
        for i in range(0, Q, 4096):
            p_, n_ = cp.build(i, min(4096, Q - i))
            ps.append(net.gate(net.mups(p_, n_))[0].double().cpu().numpy())
        logp[mode] = np.log(np.concatenate(ps))
        del net
        torch.cuda.empty_cache()
    d = logp["f16"] - logp["f16x3"]                      # per-query common mode + per-expert error
    d = d - d.mean(1, keepdims=True)                     # remove the common mode
    a = logp["f16"].argmax(1)
    E = d.shape[1]
    # pair errors against the f16 arg-max, as the gate measures them
    err = d[np.arange(Q), a][:, None] - d                # (l16_a - l16_e) - (lx3_a - lx3_e)
    mask = np.ones_like(err, bool)
    mask[np.arange(Q), a] = False
    c_half, c_all = d[:Q // 2].mean(0), d.mean(0)        # per-expert offsets estimated on the first half / on everything
    res = {"cloud": name, "queries": Q, "per_expert_offset": c_all.round(5).tolist(),
           "per_expert_error_std": d.std(0).round(5).tolist(),
           "sigma_pairs": float(np.sqrt((err[mask] ** 2).mean())), "max_pair_err": float(np.abs(err[mask]).max())}
    for tag, c, sl in (("corrected_with_first_half_offsets_second_half", c_half, slice(Q // 2, Q)), ("corrected_in_sample", c_all, slice(0, Q))):
        dc = d[sl] - c[None, :]
        ac = (logp["f16"][sl] - c[None, :]).argmax(1)
        n = dc.shape[0]
        e2 = dc[np.arange(n), ac][:, None] - dc
        m2 = np.ones_like(e2, bool)
        m2[np.arange(n), ac] = False
        res[tag] = {"sigma_pairs": float(np.sqrt((e2[m2] ** 2).mean())), "max_pair_err": float(np.abs(e2[m2]).max())}
    e_half = err[Q // 2:][mask[Q // 2:]]
    res["uncorrected_second_half"] = {"sigma_pairs": float(np.sqrt((e_half ** 2).mean())), "max_pair_err": float(np.abs(e_half).max())}
    print(json.dumps(res), flush=True)
    out.append(res)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/exp_gate_bias.txt", "w").write("\n".join(json.dumps(x) for x in out) + "\n")
