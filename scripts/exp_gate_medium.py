"""Measure-first for a THREE-stage gate: would a "medium" gating pass -- pair layout, 1x1x1 / FC layers three-product, the k^3 tap
layers single-product (nesti_model_set_gate_mix) -- sit between the plain-f16 filter and the full f16x3 gate?

For every query of the bench cloud: logits (log of the probabilities; only differences matter) of the full f16x3 gate, the
medium gate and the plain f16 gate.  Error of an approximate gate on the quantity the margin guards:
err = max_k |(l_a - l_k)_approx - (l_a - l_k)_full|, a = the approximate gate's arg-max.  Reports sigma / max of both
errors, the fraction of queries each threshold 1.5 x max would send on, the time of each pass, and the projected cost of
filter -> medium -> full against today's filter -> full.  -> gpurun_out/gate_medium.txt
The "xw" column (the numerics of a filter whose one-tap layers use the exact weights -- what round 5 then built as conv_igemm_kernel's
X2 loop) needs a measurement build of the library: make -C nesti-net_amd/csrc clean all EXTRA_CXXFLAGS=-DNESTI_EXPERIMENT_XW."""
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

from nesti_net_amd import _lib  # noqa: E402
_lib.load().nesti_experiment_mix_enable(1)      # models created below also carry the single-product copies of their tap layers
dev = torch.device("cuda:0")
cfg = NestiConfig()
N = int(os.environ.get("MED_POINTS", "100000"))
B = 25000
pts = synth.make_cloud("ellipsoid", n=N, seed=1234)[0]
cp = CloudPatches(pts, cfg, device=dev)
sp, sn = cp.build(0, 512)
W = calibrate_gate(cfg, weights.synthetic_weights(cfg), sp, sn, device=dev)
del sp, sn
net3 = NestiNet(cfg, W, dtype="f16x3", device=dev, max_batch=B)
net1 = NestiNet(cfg, W, dtype="f16", device=dev, max_batch=B)
out = {"full": [], "medium": [], "xw": [], "f16": []}
ms = {"full": 0.0, "medium": 0.0, "xw": 0.0, "f16": 0.0}


def timed(fn):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    torch.cuda.synchronize()
    return r, e0.elapsed_time(e1)


for done in range(0, N, B):
    take = min(B, N - done)
    p, n = cp.build(done, take)
    m3, m1 = net3.mups(p, n), net1.mups(p, n)
    # "xw": every activation rounded to f16, tap layers single-product, 1x1x1 / FC layers hi * (W_hi + W_lo): the NUMERICS of a plain-f16
    # filter whose 1x1x1 / FC layers use the exact weights (its time here is the emulation's, not that filter's)
    for key, net, mups, mix in (("full", net3, m3, 0), ("medium", net3, m3, 1), ("xw", net3, m3, 2), ("f16", net1, m1, 0)):
        if net is net3:
            net3.set_gate_mix(mix)
        (probs, _), t = timed(lambda: net.gate(mups))
        ms[key] += t
        out[key].append(probs.double().cpu().numpy())
    net3.set_gate_mix(0)
    del m3, m1, p, n
L = {k: np.log(np.maximum(np.concatenate(v), 1e-300)) for k, v in out.items()}
full = L["full"]
srt = np.sort(full, axis=1)
margin_full = srt[:, -1] - srt[:, -2]
lines = ["three-stage gate experiment (scripts/exp_gate_medium.py): %d queries, calibrated gate" % N,
         "gate pass per %d queries: full f16x3 %.1f ms, medium (taps single-product, 1x1 / FC three-product) %.1f ms, plain f16 %.1f ms"
         % (N, ms["full"], ms["medium"], ms["f16"])]
res = {"queries": N, "ms": ms}
for key in ("f16", "xw", "medium"):
    l = L[key]
    a = np.argmax(l, axis=1)
    rows = np.arange(N)
    d_apx = l[rows, a][:, None] - l
    d_full = full[rows, a][:, None] - full
    pe = np.abs(d_apx - d_full)
    err = pe.max(axis=1)
    sigma = float(np.sqrt((pe ** 2).sum() / (N * (cfg.n_experts - 1))))
    s2 = np.sort(l, axis=1)
    margin = s2[:, -1] - s2[:, -2]
    thr = 1.5 * err.max()
    flips = int((a != np.argmax(full, axis=1)).sum())
    res[key] = {"sigma": sigma, "max_err": float(err.max()), "p999_err": float(np.quantile(err, 0.999)), "threshold_1.5x_max": float(thr),
                "frac_below_threshold": float((margin < thr).mean()), "argmax_flips_vs_full": flips,
                "flips_above_threshold": int(((a != np.argmax(full, axis=1)) & (margin >= thr)).sum())}
    lines.append("%-7s error on a logit difference: sigma %.4g  p99.9 %.4g  max %.4g   arg-max flips vs full %d   margin < 1.5 x max: %.2f %% of the queries (flips above it: %d)"
                 % (key, sigma, res[key]["p999_err"], err.max(), flips, 100 * res[key]["frac_below_threshold"], res[key]["flips_above_threshold"]))
# stage 2 sees only the rows stage 1 flags: the medium margin of those rows against the medium threshold
l16, lm = L["f16"], L["medium"]
s16 = np.sort(l16, axis=1)
flag1 = (s16[:, -1] - s16[:, -2]) < res["f16"]["threshold_1.5x_max"]
sm = np.sort(lm, axis=1)
flag2 = flag1 & ((sm[:, -1] - sm[:, -2]) < res["medium"]["threshold_1.5x_max"])
f1, f2 = float(flag1.mean()), float(flag2.mean())
now = f1 * ms["full"]
three = f1 * ms["medium"] + f2 * ms["full"]
res["projection"] = {"stage1_flagged": f1, "stage2_flagged": f2, "recheck_ms_now": now, "recheck_ms_three_stage": three}
lines.append("exact-weight filter (xw): sigma %.4g vs the plain filter's %.4g; 1.5 x max -> %.2f %% of the queries rechecked instead of %.2f %%: "
             "recheck %.1f -> %.1f ms per %d queries if the filter's 1x1x1 / FC layers used the exact weights"
             % (res["xw"]["sigma"], res["f16"]["sigma"], 100 * res["xw"]["frac_below_threshold"], 100 * res["f16"]["frac_below_threshold"],
                res["f16"]["frac_below_threshold"] * ms["full"], res["xw"]["frac_below_threshold"] * ms["full"], N))
lines.append("filter flags %.2f %%; of those the medium gate leaves %.2f %% of all queries for the full gate" % (100 * f1, 100 * f2))
lines.append("recheck cost per %d queries: today %.1f ms (flagged x full)  ->  three-stage %.1f ms (flagged x medium + still-flagged x full): %+.1f ms"
             % (N, now, three, three - now))
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/gate_medium.txt", "w").write("\n".join(lines) + "\n")
json.dump(res, open("gpurun_out/gate_medium.json", "w"), indent=1)
print("\n".join(lines))
