"""Round-3 error attribution, on the bench's 100k cloud with the bench's calibrated synthetic weights.

1. f32 / f16 / f16x3 on all 100 000 queries -> gpurun_out/attr_probs.npz (probs, expert, normals per mode): the data the
   gate-cascade margin is derived from (f16-vs-exact logit error against the top-2 margin).
2. Sweep: the f16x3 model with ONE layer group at a time computed as plain f16 (NESTI_X3_PLAIN, model.hip: the lo * W_hi
   and hi * W_lo weight planes of the matching layers are packed as zeros), 10 240 strided queries against the f32 mode.
   NOTE: the committed results (profiles/r03_attribution_sweep.txt) were produced at commit 6f246d7, whose pair modes ran
   three activation planes [hi | lo | hi] against weights [W_hi ; W_hi ; W_lo] -- there each product has its own weight
   plane and can be switched off.  The two-plane layout of later commits only keeps the hi * W_lo switch (W_lo packed as
   zeros), so on those this script measures the weight-rounding share alone (":1" / ":2" suffixes are ignored).
   Since round 4 the NESTI_X3_PLAIN hook only exists in builds made with -DNESTI_ATTRIBUTION (the product library has no such
   switch): `make -C nesti-net_amd/csrc clean all EXTRA_CXXFLAGS=-DNESTI_ATTRIBUTION` (or a copy of the library via NESTI_LIB).
3. Gate / experts time split in f16 and f16x3 on 32 768 queries.
Prints JSON lines; -> gpurun_out/attr_sweep.txt"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import parity, synth, weights  # noqa: E402
from nesti_net_amd.calibrate import calibrate_gate  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.model import NestiNet  # noqa: E402
from nesti_net_amd.pipeline import NormalEstimator  # noqa: E402
from nesti_net_amd.provider import CloudPatches  # noqa: E402

dev = torch.device("cuda:0")
cfg = NestiConfig()
N = 100000
pts = synth.make_cloud("ellipsoid", n=N, seed=1234)[0]
cp = CloudPatches(pts, cfg, device=dev)
sp, sn = cp.build(0, 512)
W = calibrate_gate(cfg, weights.synthetic_weights(cfg), sp, sn, device=dev)
del sp, sn
os.makedirs("gpurun_out", exist_ok=True)
lines = []


def emit(d):
    print(json.dumps(d), flush=True)
    lines.append(d)


def run(dtype, pidx=None, batch=8192):
    est = NormalEstimator(cfg, W, dtype=dtype, device=dev, batch=batch)
    out = est.estimate(pts, pidx=pidx)
    del est
    torch.cuda.empty_cache()
    return out


# ---- 1. full-cloud outputs per mode (ATTR_FULL=1) --------------------------------------------------------------------
if os.environ.get("ATTR_FULL") == "1":
    full = {}
    for dt, b in (("f32", 8192), ("f16", 50000), ("f16x3", 25000)):
        t = time.time()
        full[dt] = run(dt, batch=b)
        emit({"full": dt, "seconds": time.time() - t})
    np.savez_compressed("gpurun_out/attr_probs.npz", **{"%s_%s" % (dt, k): v for dt, o in full.items()
                                                       for k, v in zip(("normals", "expert", "probs"), o)})
    for dt in ("f16", "f16x3"):
        rep = parity.compare(full[dt], full["f32"])
        emit({"full_parity": dt, **{k: rep[k] for k in ("argmax_flips", "flip_margin_max", "prob_abs_err_max")},
              "one_minus_cos": rep["one_minus_cos"]})

# ---- 2. one layer group at a time in plain f16 -----------------------------------------------------------------------
# (the reference runs on the same pidx list: a patch row's subsample key is its position in the list)
q = np.arange(3, N, N // 10240)[:10240]
ref = run("f32", pidx=q, batch=4096)
groups = [("none", "")]
for blk in (1, 2, 3, 5, 6, 8):
    for cv in (1, 2, 3):
        groups.append(("gate i%d conv%d" % (blk, cv), "inception%dgating_conv_conv%d" % (blk, cv)))
for f in (1, 2, 3, 4):
    groups.append(("gate fc%d" % f, "fc%dnoise" % f))
for blk in (1, 2, 4, 6):
    for cv in (1, 2, 3):
        groups.append(("experts i%d conv%d" % (blk, cv), "inception%dExpert_._conv%d" % (blk, cv)))
for f in (1, 2, 3, 4):
    groups.append(("experts fc%d" % f, "fc%dExpert" % f))
groups += [("all 5^3 layers", "inception[123]gating_conv_conv3,inception[12]Expert_._conv3"),
           ("all 3^3 layers @8", "inception[123]gating_conv_conv2,inception[12]Expert_._conv2"),
           ("all gate", "gating_conv,noise"), ("all experts", "Expert"),
           ("all: drop lo*W_hi only", ".:1"), ("all: drop hi*W_lo only", ".:2"), ("all plain", ".")]
for name, spec in groups:
    os.environ["NESTI_X3_PLAIN"] = spec
    rep = parity.compare(run("f16x3", pidx=q, batch=10240), ref)
    emit({"plain": name, "spec": spec, "argmax_flips": rep["argmax_flips"], "flip_margin_max": rep["flip_margin_max"],
          "prob_abs_err_max": rep["prob_abs_err_max"], "omc_p50": rep["one_minus_cos"]["p50"],
          "omc_p99": rep["one_minus_cos"]["p99"], "omc_max": rep["one_minus_cos"]["max"]})
os.environ["NESTI_X3_PLAIN"] = ""

# ---- 3. gate / experts time split ------------------------------------------------------------------------------------
B = 32768
points, n_eff = cp.build(0, B)
for dt in ("f16", "f16x3"):
    net = NestiNet(cfg, W, dtype=dt, device=dev, max_batch=B)
    mups = net.mups(points, n_eff)
    probs, expert = net.gate(mups)
    net.experts(mups, expert)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    probs, expert = net.gate(mups)
    ev[1].record()
    net.experts(mups, expert)
    ev[2].record()
    torch.cuda.synchronize()
    emit({"split": dt, "batch": B, "gate_ms": ev[0].elapsed_time(ev[1]), "experts_ms": ev[1].elapsed_time(ev[2]),
          "workspace_MB_per_query": net.lib.nesti_workspace_bytes(net._handle, B) / B / 1e6})
    del net, mups
    torch.cuda.empty_cache()
open("gpurun_out/attr_sweep.txt", "w").write("\n".join(json.dumps(x) for x in lines) + "\n")
