import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa
from nesti_net_amd import synth, weights
from nesti_net_amd.config import NestiConfig
from nesti_net_amd.pipeline import NormalEstimator
batch, part = int(sys.argv[1]), int(sys.argv[2])
cfg = NestiConfig(); W = weights.synthetic_weights(cfg)
est = NormalEstimator(cfg, W, dtype="f16", batch=batch, use_graph=True)
pts = synth.make_cloud("ellipsoid", n=30000, seed=5)[0]
full = est.prepare(pts, pidx=np.arange(batch))
partial = est.prepare(pts, pidx=np.arange(part))
def step(name, cloud):
    out = est.run(cloud); torch.cuda.synchronize(); print(name, "ok", float(out[0].abs().sum()), flush=True)
step("full(capture+replay)", full)
step("full(replay)", full)
step("partial(eager)", partial)
step("full(replay after eager)", full)
step("partial(eager)", partial)
