"""Same-box A/B of the EXPERT towers (the twin of ab_gate.py): each configuration "ENV=a,ENV2=b:dtype[:x8mask]" runs in its own
process (NESTI_LIB picks another build of the library), B random MuPS rows routed evenly over the seven experts, `reps` timed
passes of nesti_experts_forward (hipEvents around the call + the library's per-class events).

    python scripts/ab_expert.py 28672 3 "NESTI_LIB=/root/repo/.ab/libnesti_old.so:f16x8" ":f16x8" ":f16x3" ":f16x8:15"
-> one line per configuration and round (two rounds, interleaved), also appended to gpurun_out/ab_expert.txt"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(B, reps, dtype, mask):
    import torch
    sys.path.insert(0, REPO)
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import _lib, weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet
    cfg = NestiConfig()
    net = NestiNet(cfg, weights.synthetic_weights(cfg), dtype=dtype, max_batch=B)
    if mask is not None:
        net.set_x8_layers(mask)
    torch.manual_seed(0)
    lib = _lib.load()
    v = torch.rand(B, 8, 8, 8, 64, device="cuda") * 0.1
    v[..., 60:] = 0
    hi = v.to(torch.float16)
    mups = torch.cat([hi, (v - hi.float()).to(torch.float16)], dim=-1).contiguous()
    expert = (torch.arange(B, device="cuda") % cfg.n_experts).to(torch.int32)
    net.experts(mups, expert)
    torch.cuda.synchronize()
    lib.nesti_profile_enable(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = net.experts(mups, expert)
    e1.record()
    torch.cuda.synchronize()
    ms, _ = _lib.profile_read(lib)
    lib.nesti_profile_enable(0)
    print(json.dumps({"ms_per_pass": e0.elapsed_time(e1) / reps, "by_class": {c: round(ms["experts"][c] / reps, 2) for c in _lib.PROF_CONV},
                      "checksum": float(out.double().abs().sum())}))


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] != "" else None)
        sys.exit(0)
    B, reps, specs = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3:]
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    log = open(os.path.join(REPO, "gpurun_out", "ab_expert.txt"), "a")
    for rnd in range(2):
        for spec in specs:
            parts = spec.split(":")
            envs, dtype, mask = parts[0], parts[1], (parts[2] if len(parts) > 2 else "")
            env = dict(os.environ)
            for kv in filter(None, envs.split(",")):
                k, v = kv.split("=", 1)
                env[k] = v
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(B), str(reps), dtype, mask], env=env,
                               capture_output=True, text=True, timeout=900)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            msg = "round %d  %-60s %s" % (rnd, spec, line[-1] if line else "FAILED: " + r.stderr[-400:])
            print(msg, flush=True)
            log.write(msg + "\n")
