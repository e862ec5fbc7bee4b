"""Find out-of-bounds writes of nesti_forward: workspace and outputs are carved out of guarded buffers."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa
from nesti_net_amd import weights, _lib
from nesti_net_amd.config import NestiConfig
from nesti_net_amd.model import NestiNet
cfg = NestiConfig(); W = weights.synthetic_weights(cfg)
dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
net = NestiNet(cfg, W, dtype=dtype, max_batch=16)
G = 1 << 20
for B in [int(x) for x in sys.argv[2:]] or [155, 712, 1000, 4096]:
    nbytes = net.lib.nesti_workspace_bytes(net._handle, B)
    big = torch.full((nbytes + 2 * G,), 0x5A, dtype=torch.uint8, device="cuda")
    ws = big[G:G + nbytes]
    outbuf = torch.full((B * 11 * 4 + 2 * G,), 0x5A, dtype=torch.uint8, device="cuda")
    o = outbuf[G:G + B * 11 * 4].view(torch.float32)
    normals, expert, probs = o[:3 * B].view(B, 3), o[3 * B:4 * B].view(torch.int32), o[4 * B:11 * B].view(B, 7)
    pts = torch.randn(B, 1536, 3, device="cuda") * 0.3
    n_eff = torch.randint(1, 513, (B, 3), dtype=torch.int32, device="cuda")
    net.forward(pts, n_eff, out=(normals, expert, probs), ws=ws)
    torch.cuda.synchronize()
    bad = []
    for name, buf, n in (("ws", big, nbytes), ("out", outbuf, B * 11 * 4)):
        lo, hi = buf[:G], buf[G + n:]
        if not bool((lo == 0x5A).all()): bad.append(name + "-below@%d" % int((lo != 0x5A).nonzero()[-1]))
        if not bool((hi == 0x5A).all()): bad.append(name + "-above@+%d..+%d" % (int((hi != 0x5A).nonzero()[0]), int((hi != 0x5A).nonzero()[-1])))
    print("B=%d ws=%d bytes:" % (B, nbytes), "CLEAN" if not bad else bad, flush=True)
