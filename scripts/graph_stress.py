"""Stress the hipGraph replay path: many full-batch replays, optional partial (eager) batches in between."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import synth, weights  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.pipeline import NormalEstimator  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 3 * batch
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dtype = sys.argv[4] if len(sys.argv) > 4 else "f16"
cfg = NestiConfig()
W = weights.synthetic_weights(cfg)
est = NormalEstimator(cfg, W, dtype=dtype, batch=batch, use_graph=True)
pts = synth.make_cloud("ellipsoid", n=max(n_pts, 20000), seed=5)[0]
cloud = est.prepare(pts, pidx=np.arange(n_pts))
done = 0
for r in range(reps):
    out = est.run(cloud)
    if r % 10 == 9:
        torch.cuda.synchronize()
        done = r + 1
        print("replays ok after", done, "runs of", n_pts // batch, "full batches +", n_pts % batch, "eager rows", flush=True)
torch.cuda.synchronize()
print("done", float(out[0].abs().sum()))
