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
cfg = NestiConfig()
W = weights.synthetic_weights(cfg)
net = NestiNet(cfg, W, dtype="bf16", max_batch=B)
torch.manual_seed(0)
mups = (torch.randn(B, 8, 8, 8, 64, device="cuda") * 0.05).to(torch.bfloat16)
for _ in range(reps):
    probs, expert = net.gate(mups)
torch.cuda.synchronize()
print("ok", probs[0].tolist())
