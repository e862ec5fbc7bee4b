"""The pair modes (f16x3, bf16x3; argv) against the exact-fp32 mode on clouds other than the bench's: shapes x PCPNet
noise levels x density sets (BASELINE configs 3 and 4), 8 192 strided queries each, the bench's calibrated synthetic
weights.  Prints one parity line per cloud and mode (nesti_net_amd.parity.compare) -> gpurun_out/pair_mode_sweep.txt"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import parity, synth, weights  # noqa: E402
from nesti_net_amd.calibrate import calibrate_gate  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.pipeline import NormalEstimator  # noqa: E402
from nesti_net_amd.provider import CloudPatches  # noqa: E402

dev = torch.device("cuda:0")
cfg = NestiConfig()
base = synth.make_cloud("ellipsoid", n=100000, seed=1234)[0]
cp = CloudPatches(base, cfg, device=dev)
sp, sn = cp.build(0, 512)
W = calibrate_gate(cfg, weights.synthetic_weights(cfg), sp, sn, device=dev)
del cp, sp, sn
cases = [("sphere", 0.0, None, 60000), ("torus", 0.00125, None, 100000), ("box", 0.006, None, 80000),
         ("ellipsoid", 0.012, None, 100000), ("torus", 0.0, "gradient", 70000), ("sphere", 0.012, "striped", 50000),
         ("box", 0.0, "striped", 100000), ("ellipsoid", 0.006, "gradient", 90000)]
ref_est = NormalEstimator(cfg, W, dtype="f32", device=dev, batch=4096)
modes = sys.argv[1:] or ["f16x3c", "f16x3", "bf16x3"]
ests = {m: NormalEstimator(cfg, W, dtype=m, device=dev, batch=8192) for m in modes}
if "f16x3c" in ests:        # the gate margin: calibrated ONCE, on the bench's cloud -- the other clouds then test how well it travels
    from nesti_net_amd.calibrate import calibrate_gate_margin
    cp = CloudPatches(base, cfg, device=dev)
    sp, sn = cp.build(0, 1024)
    print(json.dumps({"tau": calibrate_gate_margin(ests["f16x3c"].net, sp, sn)}), flush=True)
    del cp, sp, sn
out = []
for i, (shape, noise, dens, n) in enumerate(cases):
    pts = synth.make_cloud(shape, n=n, seed=77 + i, noise=noise, density=dens)[0]
    q = np.arange(1, n, max(1, n // 8192))[:8192]
    ref = ref_est.estimate(pts, pidx=q)
    for m in modes:
        if m == "f16x3c":
            ests[m].net.cascade_stats(reset=True)
        rep = parity.compare(ests[m].estimate(pts, pidx=q), ref)
        line = {"mode": m, "cloud": "%s n=%d noise=%g density=%s" % (shape, n, noise, dens), "experts_used": int(len(np.unique(ref[1]))),
                "argmax_flips": rep["argmax_flips"], "flips_outside_margin": rep["flips_outside_margin"],
                "flip_margin_max": rep["flip_margin_max"], "prob_abs_err_max": rep["prob_abs_err_max"],
                "one_minus_cos": {k: rep["one_minus_cos"][k] for k in ("p50", "p99", "max")}, "meets_north_star": rep["meets_north_star"]}
        if m == "f16x3c":
            line["gate_cascade"] = ests[m].net.cascade_stats()
        print(json.dumps(line), flush=True)
        out.append(line)
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/pair_mode_sweep.txt", "w").write("\n".join(json.dumps(x) for x in out) + "\n")
