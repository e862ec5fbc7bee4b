"""Round-3: the two-stage gate (dtype f16x3c) on the bench's 100k cloud -- throughput next to f16 / f16x3, parity of all
three against the exact-fp32 mode, and the gate statistics.  -> gpurun_out/cascade.txt"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import parity, synth, weights  # noqa: E402
from nesti_net_amd.calibrate import calibrate_gate, calibrate_gate_margin  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.pipeline import NormalEstimator  # noqa: E402
from nesti_net_amd.provider import CloudPatches  # noqa: E402

dev = torch.device("cuda:0")
cfg = NestiConfig()
N = 100000
pts = synth.make_cloud("ellipsoid", n=N, seed=1234)[0]
cp = CloudPatches(pts, cfg, device=dev)
sp, sn = cp.build(0, 1024)
W = calibrate_gate(cfg, weights.synthetic_weights(cfg), sp[:512], sn[:512], device=dev)
lines = []


def emit(d):
    print(json.dumps(d), flush=True)
    lines.append(d)


def timed(dtype, batch, steps=2, margin=None, safety=None):
    est = NormalEstimator(cfg, W, dtype=dtype, device=dev, batch=batch)
    info = {}
    if dtype == "f16x3c":
        if margin is None:
            info["tau_calibrated"] = calibrate_gate_margin(est.net, sp, sn, **({"safety": safety} if safety else {}))
        else:
            est.net.set_gate_margin(margin)
    cloud = est.prepare(pts)
    out = est.run(cloud)
    torch.cuda.synchronize()
    if dtype == "f16x3c":
        est.net.cascade_stats(reset=True)
    t = time.perf_counter()
    for _ in range(steps):
        cloud.build_grid()
        out = est.run(cloud)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t) / steps
    if dtype == "f16x3c":
        info["stats"] = est.net.cascade_stats()
    res = [x.cpu().numpy() for x in out]
    del est, cloud, out
    torch.cuda.empty_cache()
    return res, N / el, info


ref, rate, _ = timed("f32", 8192, steps=1)
emit({"dtype": "f32", "normals_per_s": rate})
for dtype, batch, kw in (("f16", 100000, {}), ("f16x3", 33400, {}), ("f16x3c", 50000, {}), ("f16x3c", 50000, {"margin": 0.15}),
                         ("f16x3c", 50000, {"margin": 0.10}), ("f16x3c", 33400, {"margin": 0.15}), ("f16x3c", 25000, {"margin": 0.15})):
    out, rate, info = timed(dtype, batch, **kw)
    rep = parity.compare(out, ref)
    emit({"dtype": dtype, "batch": batch, **kw, "normals_per_s": rate, **info, "argmax_flips": rep["argmax_flips"],
          "flip_margin_max": rep["flip_margin_max"], "prob_abs_err_max": rep["prob_abs_err_max"],
          "omc_max": rep["one_minus_cos"]["max"], "omc_max_incl_flips": rep["one_minus_cos"]["max_incl_flips"]})
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/cascade.txt", "w").write("\n".join(json.dumps(x) for x in lines) + "\n")
