"""BASELINE config 5 (32 clouds in flight, 50k-98k points, gradient / striped density sets and all PCPNet noise levels) through the
headline mode (f16x8c since round 6; STREAM32_MODE=f16x3c for round 5's) and through f16x3 (every query decided by the pair-mode gate,
every multiply three f16 products): the expert arg-max must be identical on every query of every cloud, the normals identical
(f16x3c) or within 1 - cos <= 2.5e-6 (f16x8c: the FP8 cross terms of the experts' 5^3 layers), and the two-stage gate's counters show
how the margin behaved over the 2.4 M queries of one pass (tau_eff / max_margin_err >= 1.5 by construction; rounds_widened = forward
calls whose widening round was not empty).
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
HEAD = os.environ.get("STREAM32_MODE", "f16x8c")
for mode, batch in ((HEAD, 100000), ("f16x3", 50000)):
    est = NormalEstimator(cfg, W, dtype=mode, device=dev, batch=batch)
    clouds = [est.prepare(p) for p, _ in clouds_np]
    if mode == HEAD:
        sp, sn = clouds[0].build(0, 1024)
        res["tau"] = calibrate_gate_margin(est.net, sp, sn)
        if mode in ("f16x8", "f16x8c"):
            from nesti_net_amd.calibrate import calibrate_x8_guard
            if os.environ.get("STREAM32_X8_LAYERS"):
                est.net.set_x8_layers(int(os.environ["STREAM32_X8_LAYERS"], 0))
            res["x8_guard_thr"] = calibrate_x8_guard(est.net, sp, sn)
        del sp, sn
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = est.run_many([(c, 0, c.patch_count) for c in clouds])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    experts[mode] = np.concatenate([o[1].cpu().numpy() for o in outs])
    normals = np.concatenate([o[0].cpu().numpy() for o in outs])
    res[mode] = {"queries": int(len(experts[mode])), "seconds": el, "normals_per_s": len(experts[mode]) / el}
    if mode == HEAD:
        st = est.net.cascade_stats()
        st["tau_eff_over_max_margin_err"] = st["tau_eff"] / st["max_margin_err"] if st["max_margin_err"] else None
        st["tau_over_max_margin_err"] = st["tau"] / st["max_margin_err"] if st["max_margin_err"] else None
        res["gate_cascade"] = st
        if mode in ("f16x8", "f16x8c"):
            res["x8_guard"] = est.net.x8_guard_stats()
        n_c = normals
    else:
        same = experts[HEAD] == experts["f16x3"]
        res["normals_bitwise_equal_where_argmax_agrees"] = bool(np.array_equal(n_c[same], normals[same]))
        a, b = n_c[same].astype(np.float64), normals[same].astype(np.float64)
        omc = 1.0 - (a * b).sum(1) / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-300)
        res["one_minus_cos_vs_f16x3"] = {"p50": float(np.quantile(omc, .5)), "p99": float(np.quantile(omc, .99)), "p9999": float(np.quantile(omc, .9999)),
                                         "max": float(omc.max()), "over_2.5e-6": int((omc > 2.5e-6).sum()), "over_1e-5": int((omc > 1e-5).sum())}
    del est, clouds, outs
    torch.cuda.empty_cache()
res["headline_mode"] = HEAD
res["argmax_differences_headline_vs_f16x3"] = int((experts[HEAD] != experts["f16x3"]).sum())
res["routing_histogram"] = np.bincount(experts["f16x3"], minlength=7).tolist()
print(json.dumps(res))
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", "stream32_check.json"), "w").write(json.dumps(res) + "\n")
