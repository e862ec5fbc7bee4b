"""BASELINE config 5 (32 clouds in flight, 50k-98k points, gradient / striped density sets and all PCPNet noise levels) through the
headline mode f16x3c and through f16x3 (every query decided by the pair-mode gate): the expert arg-max must be identical on every
query of every cloud, and the two-stage gate's counters show how the margin behaved over the 2.4 M queries of one pass
(tau_eff / max_margin_err >= 1.5 by construction; rounds_widened = forward calls whose widening round was not empty).
-> gpurun_out/stream32_check.json"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import weights  # noqa: E402
from nesti_net_amd.calibrate import calibrate_gate, calibrate_gate_margin  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.pipeline import NormalEstimator  # noqa: E402
from nesti_net_amd.provider import CloudPatches  # noqa: E402

dev = torch.device("cuda:0")
cfg = NestiConfig()
clouds_np = bench.make_clouds(32, 100000, stream=True)
cp = CloudPatches(clouds_np[0][0], cfg, device=dev)
sp, sn = cp.build(0, 512)
W = calibrate_gate(cfg, weights.synthetic_weights(cfg), sp, sn, device=dev)
del cp, sp, sn
res = {}
experts = {}
for mode, batch in (("f16x3c", 100000), ("f16x3", 50000)):
    est = NormalEstimator(cfg, W, dtype=mode, device=dev, batch=batch)
    clouds = [est.prepare(p) for p, _ in clouds_np]
    if mode == "f16x3c":
        sp, sn = clouds[0].build(0, 1024)
        res["tau"] = calibrate_gate_margin(est.net, sp, sn)
        del sp, sn
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = est.run_many([(c, 0, c.patch_count) for c in clouds])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    experts[mode] = np.concatenate([o[1].cpu().numpy() for o in outs])
    normals = np.concatenate([o[0].cpu().numpy() for o in outs])
    res[mode] = {"queries": int(len(experts[mode])), "seconds": el, "normals_per_s": len(experts[mode]) / el}
    if mode == "f16x3c":
        st = est.net.cascade_stats()
        st["tau_eff_over_max_margin_err"] = st["tau_eff"] / st["max_margin_err"] if st["max_margin_err"] else None
        st["tau_over_max_margin_err"] = st["tau"] / st["max_margin_err"] if st["max_margin_err"] else None
        res["gate_cascade"] = st
        n_c = normals
    else:
        res["normals_bitwise_equal_where_argmax_agrees"] = bool(np.array_equal(n_c[experts["f16x3c"] == experts["f16x3"]],
                                                                               normals[experts["f16x3c"] == experts["f16x3"]]))
    del est, clouds, outs
    torch.cuda.empty_cache()
res["argmax_differences_f16x3c_vs_f16x3"] = int((experts["f16x3c"] != experts["f16x3"]).sum())
res["routing_histogram"] = np.bincount(experts["f16x3"], minlength=7).tolist()
print(json.dumps(res))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", "stream32_check.json"), "w").write(json.dumps(res) + "\n")
