"""How long the host takes to ENQUEUE one 100k-query step (two 50 176-query calls on two streams) against the step itself: the launch
count grew with the conditioning guard and the widening passes."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nesti_net_amd
from nesti_net_amd import synth, weights
from nesti_net_amd.calibrate import calibrate_gate, calibrate_gate_margin, calibrate_x8_guard
from nesti_net_amd.config import NestiConfig
from nesti_net_amd.pipeline import NormalEstimator
from nesti_net_amd.provider import CloudPatches
dev=torch.device("cuda:0"); cfg=NestiConfig()
pts=synth.make_cloud("ellipsoid", n=100000, seed=1234)[0]
cp=CloudPatches(pts,cfg,device=dev); sp,sn=cp.build(0,512)
W=calibrate_gate(cfg, weights.synthetic_weights(cfg), sp, sn, device=dev)
for dtype in ("f16x8c","f16x3c"):
    est=NormalEstimator(cfg,W,dtype=dtype,device=dev,batch=50176,n_streams=2)
    cloud=est.prepare(pts); sp,sn=cloud.build(0,1024); calibrate_gate_margin(est.net,sp,sn)
    if dtype=="f16x8c": calibrate_x8_guard(est.net,sp,sn)
    est.run(cloud); torch.cuda.synchronize()
    for rep in range(2):
        t0=time.perf_counter(); est.run(cloud); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        print(dtype, "host enqueue %.1f ms, total %.1f ms" % (1e3*(t1-t0), 1e3*(t2-t0)), flush=True)
    del est, cloud
    torch.cuda.empty_cache()
