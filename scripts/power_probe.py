"""Is the conv stack clock/power bound?  Times the gating tower on random vs all-zero weights+input
(identical instruction stream, different data toggling)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import weights  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.model import NestiNet  # noqa: E402

B = 4096
cfg = NestiConfig()
W = weights.synthetic_weights(cfg)
for zero in (False, True, False, True):
    Wz = {k: (np.zeros_like(v) if zero and not k.endswith("/bn/var") else v) for k, v in W.items()}
    net = NestiNet(cfg, Wz, dtype="bf16", max_batch=B)
    torch.manual_seed(0)
    mups = (torch.randn(B, 8, 8, 8, 64, device="cuda") * (0.0 if zero else 0.05)).to(torch.bfloat16)
    for _ in range(2):
        net.gate(mups)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        net.gate(mups)
    torch.cuda.synchronize()
    print("zero" if zero else "random", "gate ms per 4096-query batch: %.2f" % ((time.perf_counter() - t) / 5 * 1e3))
    del net
