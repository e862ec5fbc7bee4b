"""Profiling driver: the whole network (MuPS + gate + routed experts) on one batch of synthetic patches
(used under rocprofv3 --kernel-trace to get per-layer launch durations)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import weights  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.model import NestiNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 25000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = NestiConfig()
W = weights.synthetic_weights(cfg)
net = NestiNet(cfg, W, dtype="bf16", max_batch=B)
torch.manual_seed(0)
S, P = len(cfg.patch_radius), cfg.num_point
pts = torch.randn(B, S * P, 3, device="cuda") * 0.3
n_eff = torch.full((B, S), P, dtype=torch.int32, device="cuda")
for _ in range(reps):
    normals, expert, probs = net(pts, n_eff)
torch.cuda.synchronize()
print("ok", torch.bincount(expert.long(), minlength=cfg.n_experts).tolist())
