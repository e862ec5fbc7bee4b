"""Child process of tests/test_gpu_dist.py: one rank of a world-size-2 job on ONE GPU (gloo, cuda:0) running the real
sharded estimator; writes its gathered result to <out>.rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import dist as nd  # noqa: E402
from nesti_net_amd import synth, weights  # noqa: E402
from nesti_net_amd.calibrate import calibrate_gate  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.pipeline import NormalEstimator  # noqa: E402
from nesti_net_amd.provider import CloudPatches  # noqa: E402


def build(model, dev):
    cfg = NestiConfig.for_model(model)
    W = weights.synthetic_weights(cfg)
    pts = synth.make_cloud("torus", n=5001, seed=77, noise=0.00125)[0]         # odd size: ragged shards
    if model == "experts_n_est":
        cp = CloudPatches(pts, cfg, device=dev)
        sp, sn = cp.build(0, 512)
        W = calibrate_gate(cfg, W, sp, sn, device=dev)
    return cfg, W, pts


def main():
    out, model = sys.argv[1], sys.argv[2]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    cfg, W, pts = build(model, dev)
    est = NormalEstimator(cfg, W, dtype="f16", device=dev, batch=1024)
    cloud = est.prepare(pts)
    normals, expert, probs = nd.estimate_sharded(est, cloud)
    torch.cuda.synchronize()
    rank = dist.get_rank() if world > 1 else 0
    np.savez(out + ".rank%d.npz" % rank, normals=normals.cpu().numpy(),
             expert=np.zeros(0) if expert is None else expert.cpu().numpy(),
             probs=np.zeros(0) if probs is None else probs.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
