"""Profiling driver: gating tower only, one batch of random MuPS input (used under rocprofv3)."""
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
dtype = sys.argv[3] if len(sys.argv) > 3 else "f16"
cfg = NestiConfig()
W = weights.synthetic_weights(cfg)
net = NestiNet(cfg, W, dtype=dtype, max_batch=B)
if dtype == "f16x3c":
    net.set_gate_margin(0.0)            # the filter pass alone (nothing is rechecked)
torch.manual_seed(0)
v = torch.randn(B, 8, 8, 8, 64, device="cuda") * 0.05
v[..., 60:] = 0
tdt = torch.bfloat16 if dtype.startswith("bf16") else torch.float16
if net.mups_cstride == 64:
    mups = v.to(tdt)
else:                                   # pair layout [hi | lo] per 64-channel group
    hi = v.to(tdt)
    mups = torch.cat([hi, (v - hi.float()).to(tdt)], dim=-1).contiguous()
for _ in range(reps):
    probs, expert = net.gate(mups)
torch.cuda.synchronize()
print("ok", probs[0].tolist())
