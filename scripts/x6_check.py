"""The FP6 (e2m3, block-scaled) form of the experts' cross terms against the FP8 form and against f16x3, guard OFF, on 10 000 queries of
the bench's cloud shape with a calibrated gate (the numbers scripts/exp_fp8_cross.py predicted: |dn| p50 8.6e-5 against e4m3's 7.4e-5
with all four tap layers at 8^3), per layer mask; then same-process timing of the experts in the three forms.
-> gpurun_out/x6_check.txt"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import synth, weights  # noqa: E402
from nesti_net_amd.calibrate import calibrate_gate  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.model import NestiNet  # noqa: E402
from nesti_net_amd.provider import CloudPatches  # noqa: E402


def omc(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return 1.0 - (a * b).sum(1) / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-300)


dev = torch.device("cuda:0")
cfg = NestiConfig()
N, Q = 100000, int(os.environ.get("X6_QUERIES", "10000"))
pts = synth.make_cloud("ellipsoid", n=N, seed=1234)[0]
q = np.arange(0, N, N // Q)[:Q]
cp = CloudPatches(pts, cfg, device=dev, pidx=q)
p_d, n_d = cp.build(0, Q)
W = calibrate_gate(cfg, weights.synthetic_weights(cfg), p_d[:512], n_d[:512], device=dev)
n3, e3, p3 = NestiNet(cfg, W, dtype="f16x3", device=dev, max_batch=Q)(p_d, n_d)
n3, e3 = n3.cpu().numpy(), e3.cpu().numpy()
net = NestiNet(cfg, W, dtype="f16x8", device=dev, max_batch=Q)
net.set_x8_guard(-1.0)                     # guard off: the raw residual of each form
lines = []
for fmt in (8, 6):
    net.set_x8_format(fmt)
    for mask in (0xF, 0xA, 0x5):
        net.set_x8_layers(mask)
        n8, e8, p8 = net(p_d, n_d)
        torch.cuda.synchronize()
        assert np.array_equal(e8.cpu().numpy(), e3) and torch.equal(p8, p3)
        o = omc(n8.cpu().numpy(), n3)
        dn = np.linalg.norm(n8.cpu().numpy().astype(np.float64) - n3, axis=1)
        lines.append("format %d mask 0x%X: 1-cos p50 %.3g p99 %.3g max %.3g   |dn| p50 %.3g p99 %.3g max %.3g   per expert max %s"
                     % (fmt, mask, np.quantile(o, .5), np.quantile(o, .99), o.max(), np.quantile(dn, .5), np.quantile(dn, .99), dn.max(),
                        " ".join("%.2g" % o[e3 == e].max() for e in range(cfg.n_experts))))
        print(lines[-1], flush=True)
net.set_x8_layers(0xF)
# timing: the experts on routed rows, the three forms in one process
mups = net.mups(p_d, n_d)
ex = torch.from_numpy(e3).to(dev)
net3 = NestiNet(cfg, W, dtype="f16x3", device=dev, max_batch=Q)
mups3 = net3.mups(p_d, n_d)
for rnd in range(2):
    for name, nn_, mm, fmt in (("f16x3", net3, mups3, None), ("e4m3", net, mups, 8), ("e2m3", net, mups, 6)):
        if fmt:
            nn_.set_x8_format(fmt)
        nn_.experts(mm, ex)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            nn_.experts(mm, ex)
        e1.record()
        torch.cuda.synchronize()
        lines.append("round %d experts on %d routed queries, %s: %.2f ms per pass" % (rnd, Q, name, e0.elapsed_time(e1) / 3))
        print(lines[-1], flush=True)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
open(os.path.join(REPO, "gpurun_out", "x6_check.txt"), "w").write("\n".join(lines) + "\n")
