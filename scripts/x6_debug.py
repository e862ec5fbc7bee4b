import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import nesti_net_amd  # noqa
from nesti_net_amd import synth, weights
from nesti_net_amd.calibrate import calibrate_gate
from nesti_net_amd.config import NestiConfig
from nesti_net_amd.model import NestiNet
from nesti_net_amd.provider import CloudPatches
dev = torch.device("cuda:0")
cfg = NestiConfig()
N, Q = 100000, 2000
pts = synth.make_cloud("ellipsoid", n=N, seed=1234)[0]
q = np.arange(0, N, N // Q)[:Q]
cp = CloudPatches(pts, cfg, device=dev, pidx=q)
p_d, n_d = cp.build(0, Q)
W = calibrate_gate(cfg, weights.synthetic_weights(cfg), p_d[:512], n_d[:512], device=dev)
n3, e3, p3 = NestiNet(cfg, W, dtype="f16x3", device=dev, max_batch=Q)(p_d, n_d)
n3 = n3.cpu().numpy().astype(np.float64)
net = NestiNet(cfg, W, dtype="f16x8", device=dev, max_batch=Q)
net.set_x8_guard(-1.0)
for fmt in (8, 6):
    net.set_x8_format(fmt)
    for mask in (1, 2, 4, 8):
        net.set_x8_layers(mask)
        n8, _, _ = net(p_d, n_d)
        dn = np.linalg.norm(n8.cpu().numpy().astype(np.float64) - n3, axis=1)
        print("fmt %d mask %d: |dn| p50 %.3g max %.3g   |n| p50 %.3g" % (fmt, mask, np.quantile(dn, .5), dn.max(), np.quantile(np.linalg.norm(n3, axis=1), .5)), flush=True)
