"""Throughput of the three subsample modes on one 100k-point cloud (all 100 000 queries; calibrated gate and margin; two passes each,
the second timed): 'hash' (the fused product path, two streams), 'reference' (round 6: the reference's own subsample order on the
GPU) and 'reference_host' (scipy + numpy on the host).  Checks that the two reference modes write identical normals.
Writes gpurun_out/reference_order.json (-> profiles/r06_reference_order.json)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import synth, weights  # noqa: E402
from nesti_net_amd.calibrate import calibrate_gate, calibrate_gate_margin  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.pipeline import NormalEstimator  # noqa: E402
from nesti_net_amd.provider import CloudPatches  # noqa: E402

dev = torch.device("cuda:0")
cfg = NestiConfig()
N = int(os.environ.get("REF_POINTS", "100000"))
dtype = os.environ.get("REF_DTYPE", "f16x8c")
pts = synth.make_cloud("ellipsoid", n=N, seed=1234)[0]
cp = CloudPatches(pts, cfg, device=dev)
sp, sn = cp.build(0, 512)
W = calibrate_gate(cfg, weights.synthetic_weights(cfg), sp, sn, device=dev)
res = {"points": N, "dtype": dtype, "host_cores": os.cpu_count()}
outs = {}
for mode in ("hash", "reference", "reference_host"):
    est = NormalEstimator(cfg, W, dtype=dtype, device=dev, batch=50176 if mode == "hash" else 25088, n_streams=2, subsample=mode)
    cloud = est.prepare(pts)
    sp, sn = cloud.build(0, 1024)
    calibrate_gate_margin(est.net, sp, sn)
    est.run(cloud)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = est.run(cloud)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    outs[mode] = [t.cpu().numpy() for t in out]
    res[mode] = {"seconds_per_cloud": dt, "normals_per_sec": N / dt, "batch": est.batch}
    if mode == "reference":
        # the two kernels of the mode on their own (hipEvents): ball sizes of all rows, reference-order rows of one batch
        from nesti_net_amd.refsample import RefStream
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        cloud.count_balls(0, N)
        ev[0].record(); sizes_d = cloud.count_balls(0, N); ev[1].record()
        sizes = sizes_d.cpu().numpy()
        nb = min(N, 25088)
        t0 = time.perf_counter()
        picks, offs = RefStream(3627473).picks(sizes[:nb].ravel(), cfg.num_point)
        t_replay = time.perf_counter() - t0
        pk, of = torch.from_numpy(picks.view(np.int16).copy()).to(dev), torch.from_numpy(offs.copy()).to(dev)
        cloud.build_reference_order(0, nb, pk, of)
        ev[2].record(); cloud.build_reference_order(0, nb, pk, of); ev[3].record()
        torch.cuda.synchronize()
        res["kernels"] = {"patches_count_kernel_ms_per_100k": ev[0].elapsed_time(ev[1]) * 1e5 / N,
                          "patches_ref_kernel_ms_per_100k": ev[2].elapsed_time(ev[3]) * 1e5 / nb,
                          "host_replay_s_per_100k": t_replay * 1e5 / nb, "picks_MB_per_100k": picks.nbytes * 1e5 / nb / 1e6}
        res["ball_sizes"] = {"max": sizes.max(0).tolist(), "mean": sizes.mean(0).round(1).tolist(),
                             "over_full_frac": (sizes > cfg.num_point).mean(0).round(4).tolist(), "sum_over_full": int(sizes[sizes > cfg.num_point].sum())}
    print(mode, res[mode], flush=True)
    del est, cloud
    torch.cuda.empty_cache()
# the two reference modes consumed their streams twice (warm-up + timed pass), identically: same rows, same normals
res["reference_equals_reference_host"] = bool(all(np.array_equal(a, b) for a, b in zip(outs["reference"], outs["reference_host"])))
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/reference_order.json", "w"), indent=1)
