"""Which expert tap layers at 8^3 can take their cross terms through FP8 inside the 2.5e-6 bar?  Real kernels, all 100 000 queries of
the bench cloud (calibrated gate, f16x3 gate decisions): per layer mask (bit 0 / 1 = inception1 conv2 (3^3) / conv3 (5^3), bit 2 / 3 =
inception2 conv2 / conv3) the routed experts' normals against f16x3's -- 1 - cos p50 / p99 / max, queries above 2.5e-6 -- and the time
of the expert pass.  -> gpurun_out/x8_masks.txt (-> profiles/r06_x8_masks.txt)"""
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

MASKS = [0x0, 0x8, 0x2, 0xA, 0x4, 0x1, 0xE, 0xB, 0xF]
dev = torch.device("cuda:0")
cfg = NestiConfig()
N, B = 100000, 25000
pts = synth.make_cloud("ellipsoid", n=N, seed=1234)[0]
cp = CloudPatches(pts, cfg, device=dev)
sp, sn = cp.build(0, 512)
W = calibrate_gate(cfg, weights.synthetic_weights(cfg), sp, sn, device=dev)
del sp, sn
net3 = NestiNet(cfg, W, dtype="f16x3", device=dev, max_batch=B)
net8 = NestiNet(cfg, W, dtype="f16x8", device=dev, max_batch=B)
ref, outs, ms = [], {m: [] for m in MASKS}, {m: 0.0 for m in MASKS}
ms3 = 0.0
for done in range(0, N, B):
    p, n = cp.build(done, B)
    mups = net3.mups(p, n)
    _, expert = net3.gate(mups)
    net3.experts(mups, expert)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = net3.experts(mups, expert); e1.record(); torch.cuda.synchronize()
    ms3 += e0.elapsed_time(e1)
    ref.append(r.double().cpu().numpy())
    for m in MASKS:
        net8.set_x8_layers(m)
        net8.experts(mups, expert)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o = net8.experts(mups, expert); e1.record(); torch.cuda.synchronize()
        ms[m] += e0.elapsed_time(e1)
        outs[m].append(o.double().cpu().numpy())
ref = np.concatenate(ref)
lines = ["expert tap layers at 8^3 through FP8, real kernels, %d queries (scripts/exp_x8_masks.py); f16x3 expert pass %.1f ms" % (N, ms3)]
names = {0: "i1 conv2 (3^3)", 1: "i1 conv3 (5^3)", 2: "i2 conv2 (3^3)", 3: "i2 conv3 (5^3)"}
for m in MASKS:
    o = np.concatenate(outs[m])
    omc = 1.0 - (o * ref).sum(1) / np.maximum(np.linalg.norm(o, axis=1) * np.linalg.norm(ref, axis=1), 1e-300)
    lines.append("mask %s  %-62s expert pass %7.1f ms   1-cos p50 %.3g p99 %.3g max %.3g   over 2.5e-6: %d   over 1e-5: %d"
                 % (format(m, "04b"), " + ".join(names[b] for b in range(4) if (m >> b) & 1) or "none (f16x3 proper)", ms[m],
                    np.quantile(omc, .5), np.quantile(omc, .99), omc.max(), int((omc > 2.5e-6).sum()), int((omc > 1e-5).sum())))
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/x8_masks.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
