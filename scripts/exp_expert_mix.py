"""VERDICT r04 item 1, measured before anything is built on it: can the experts' k^3 tap layers run ONE 16-bit product
(nesti_model_set_expert_mix) behind a device-side conditioning test that rechecks only an ill-conditioned tail in f16x3?

For every query of the bench's 100k cloud (calibrated gate, f16x3 gate decisions) the routed expert is evaluated in f16x3
(reference of this experiment: within 4.3e-7 of the f32 mode, profiles/r04_bench_n1.json) and with each mask of MASKS; per
query: |n| (norm of the raw expert output), |dn| = |n_mix - n_f16x3|, 1 - cos.  Reported per mask:
  * the 1 - cos distribution of the mixed pass alone, |dn| and |n| quantiles, time of the expert pass per 100k queries;
  * the conditioning test the verdict proposes: recheck a query when |n_mix| < K x (largest |dn| seen on this expert's
    queries), K = 1 / theta with theta = sqrt(2 x target): the rechecked fraction, and the largest 1 - cos among the queries
    it lets through, for three targets; and the same with the error bound taken from a 1024-query calibration prefix
    x 1.5 (the self-widening form the gate uses);
  * the ORACLE threshold: the smallest |n| cut that lets no query above the target through -- the best any |n|-only rule
    can do on this cloud, no safety factor;
  * expected step time = cheap pass + recheck fraction x full pass, against the full pass.
Writes gpurun_out/expert_mix.json and a text table gpurun_out/expert_mix.txt (-> profiles/r05_expert_mix.txt)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import synth, weights  # noqa: E402
from nesti_net_amd.calibrate import calibrate_gate  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402
from nesti_net_amd.model import NestiNet  # noqa: E402
from nesti_net_amd.provider import CloudPatches  # noqa: E402

MASKS = [(0b001000, "i2 conv3 (5^3 256->128)"), (0b000100, "i2 conv2 (3^3)"), (0b001100, "i2 conv2 + conv3"),
         (0b001010, "both 5^3 at 8^3"), (0b001111, "all taps at 8^3"), (0b111111, "all taps at 8^3 and 4^3")]
TARGETS = (2.5e-6, 5e-6, 1e-5)

from nesti_net_amd import _lib  # noqa: E402
_lib.load().nesti_experiment_mix_enable(1)      # models created below also carry the single-product copies of their tap layers
dev = torch.device("cuda:0")
cfg = NestiConfig()
N = int(os.environ.get("MIX_POINTS", "100000"))
B = 25000
pts = synth.make_cloud("ellipsoid", n=N, seed=1234)[0]
cp = CloudPatches(pts, cfg, device=dev)
sp, sn = cp.build(0, 512)
W = calibrate_gate(cfg, weights.synthetic_weights(cfg), sp, sn, device=dev)
del sp, sn
net = NestiNet(cfg, W, dtype="f16x3", device=dev, max_batch=B)

expert_all, ref_all = [], []
mix_all = {m: [] for m, _ in MASKS}
ms = {m: 0.0 for m, _ in MASKS}
ms[0] = 0.0
for done in range(0, N, B):
    take = min(B, N - done)
    p, n = cp.build(done, take)
    mups = net.mups(p, n)
    _, expert = net.gate(mups)
    for mask in [0] + [m for m, _ in MASKS]:
        net.set_expert_mix(mask)
        net.experts(mups, expert)                     # warm
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = net.experts(mups, expert)
        e1.record()
        torch.cuda.synchronize()
        ms[mask] += e0.elapsed_time(e1)
        (ref_all if mask == 0 else mix_all[mask]).append(out.double().cpu().numpy())
    net.set_expert_mix(0)
    expert_all.append(expert.cpu().numpy())
    del mups, p, n
expert = np.concatenate(expert_all)
ref = np.concatenate(ref_all)
nref = np.linalg.norm(ref, axis=1)
E = cfg.n_experts
lines, res = [], {"queries": N, "routing": np.bincount(expert, minlength=E).tolist(), "full_pass_ms": ms[0], "masks": []}


def q(v, x):
    return float(np.quantile(v, x)) if len(v) else 0.0


lines.append("expert-side filter experiment (scripts/exp_expert_mix.py): %d queries of the bench cloud, f16x3 gate decisions, "
             "routing %s" % (N, res["routing"]))
lines.append("full f16x3 expert pass %.1f ms; |n| of the f16x3 outputs: p1 %.3g p10 %.3g p50 %.3g p90 %.3g min %.3g"
             % (ms[0], q(nref, .01), q(nref, .1), q(nref, .5), q(nref, .9), nref.min()))
for mask, name in MASKS:
    mix = np.concatenate(mix_all[mask])
    nm = np.linalg.norm(mix, axis=1)
    dn = np.linalg.norm(mix - ref, axis=1)
    omc = 1.0 - (mix * ref).sum(1) / np.maximum(nm * nref, 1e-300)
    ent = {"mask": mask, "layers": name, "mixed_pass_ms": ms[mask],
           "one_minus_cos": {"p50": q(omc, .5), "p90": q(omc, .9), "p99": q(omc, .99), "p999": q(omc, .999), "max": float(omc.max())},
           "frac_over_1e-5": float((omc > 1e-5).mean()), "frac_over_2.5e-6": float((omc > 2.5e-6).mean()),
           "dn": {"p50": q(dn, .5), "p99": q(dn, .99), "max": float(dn.max())},
           "dn_over_n_corr": float(np.corrcoef(np.log(dn + 1e-30), np.log(nm + 1e-30))[0, 1]), "rules": []}
    lines.append("")
    lines.append("mask %s  %-28s mixed pass %.1f ms (%.0f %% of full)   1-cos p50 %.3g p99 %.3g p99.9 %.3g max %.3g   over 1e-5: %.3f %%"
                 % (format(mask, "06b"), name, ms[mask], 100 * ms[mask] / ms[0], ent["one_minus_cos"]["p50"], ent["one_minus_cos"]["p99"],
                    ent["one_minus_cos"]["p999"], ent["one_minus_cos"]["max"], 100 * ent["frac_over_1e-5"]))
    lines.append("   |dn| p50 %.3g p99 %.3g max %.3g   corr(log|dn|, log|n|) %.2f" % (ent["dn"]["p50"], ent["dn"]["p99"], ent["dn"]["max"], ent["dn_over_n_corr"]))
    for target in TARGETS:
        theta = (2 * target) ** 0.5
        # (a) threshold from the largest error over ALL of the expert's queries (hindsight), (b) from a 1024-query prefix x 1.5
        for rule, prefix in (("max_all", None), ("calib1024_x1.5", 1024)):
            recheck = np.zeros(N, bool)
            for e in range(E):
                sel = np.nonzero(expert == e)[0]
                if not len(sel):
                    continue
                cal = sel if prefix is None else sel[sel < prefix * E]          # the expert's share of the first 1024 x E queries
                bound = dn[cal].max() * (1.0 if prefix is None else 1.5) if len(cal) else np.inf
                recheck[sel] = nm[sel] < bound / theta
            through = ~recheck
            worst = float(omc[through].max()) if through.any() else 0.0
            cost = (ms[mask] + recheck.mean() * ms[0]) / ms[0]
            ent["rules"].append({"target": target, "rule": rule, "rechecked_frac": float(recheck.mean()),
                                 "max_one_minus_cos_unrechecked": worst, "holds_target": bool(worst <= target),
                                 "holds_1e-5": bool(worst <= 1e-5), "expert_time_vs_full": cost})
            lines.append("   target %.1e  %-15s recheck %5.1f %%   worst 1-cos let through %.3g (%s)   expert time %.2f x full"
                         % (target, rule, 100 * recheck.mean(), worst, "ok" if worst <= target else "OVER", cost))
        # (c) oracle |n| cut: rechecks exactly the queries at or below the largest |n| that still violates the target
        bad = omc > target
        cut = nm[bad].max() if bad.any() else 0.0
        rc = float((nm <= cut).mean())
        ent["rules"].append({"target": target, "rule": "oracle_cut", "rechecked_frac": rc, "expert_time_vs_full": (ms[mask] + rc * ms[0]) / ms[0]})
        lines.append("   target %.1e  oracle |n| cut   recheck %5.1f %%   (no safety factor: the best an |n|-only rule can do here)   expert time %.2f x full"
                     % (target, 100 * rc, (ms[mask] + rc * ms[0]) / ms[0]))
    res["masks"].append(ent)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/expert_mix.json", "w"), indent=1)
open("gpurun_out/expert_mix.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
