"""Profiling driver: EXPERT towers only (the twin of prof_gate.py; used under rocprofv3 by scripts/pmc_gate.sh with
PROF_DRIVER=prof_expert.py): one batch of random MuPS input routed half to a single-scale expert (Expert_0) and half to the
three-scale one (Expert_6), through nesti_experts_forward.  dtype f16x3 = the pair K loop everywhere, f16x8 = the FP8 cross-term
loop in the 5^3 tap layers (conv8n_kernel<2, 5, 2, .>)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import weights  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.model import NestiNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dtype = sys.argv[3] if len(sys.argv) > 3 else "f16x8"
cfg = NestiConfig()
W = weights.synthetic_weights(cfg)
net = NestiNet(cfg, W, dtype=dtype, max_batch=B)
torch.manual_seed(0)
v = torch.rand(B, 8, 8, 8, 64, device="cuda") * 0.1          # MuPS-like: non-negative, O(0.05)
v[..., 60:] = 0
hi = v.to(torch.float16)
mups = torch.cat([hi, (v - hi.float()).to(torch.float16)], dim=-1).contiguous()     # pair layout [hi | lo] per 64-channel group
expert = torch.zeros(B, dtype=torch.int32, device="cuda")
expert[B // 2:] = 6
for _ in range(reps):
    out = net.experts(mups, expert)
torch.cuda.synchronize()
print("ok", out[0].tolist(), out[-1].tolist())
