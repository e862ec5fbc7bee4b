"""Same-box kernel A/B on the gating tower (its layers are every kernel class of the path: conv8 5^3 / 3^3, the 1x1x1 |
avg-pool launches, the taps at 4^3 / 2^3, FC): each configuration "ENV=a,ENV2=b:dtype" runs in its own process (the
library reads its switches once), B random MuPS rows, `reps` timed passes with the library's per-class hipEvents.

    python scripts/ab_gate.py 32768 3 "NESTI_LIB=/root/repo/.ab/libnesti_old.so:f16" ":f16" "NESTI_LIB=/root/repo/.ab/libnesti_old.so:f16x3" ":f16x3"
-> one line per configuration and repetition round (two rounds, interleaved), also appended to gpurun_out/ab_gate.txt"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(B, reps, dtype):
    import torch
    sys.path.insert(0, REPO)
    import nesti_net_amd  # noqa: F401
    from nesti_net_amd import _lib, weights
    from nesti_net_amd.config import NestiConfig
    from nesti_net_amd.model import NestiNet, _TORCH_DT
    cfg = NestiConfig()
    W = weights.synthetic_weights(cfg)
    net = NestiNet(cfg, W, dtype=dtype, max_batch=B)
    if dtype == "f16x3c":
        net.set_gate_margin(float(os.environ.get("AB_TAU", "1e30")))
    torch.manual_seed(0)
    lib = _lib.load()
    cs = net.mups_cstride
    v = torch.randn(B, 8, 8, 8, 64, device="cuda") * 0.05
    v[..., 60:] = 0
    if cs == 64:
        mups = v.to(_TORCH_DT[dtype])
    else:                       # pair layout [hi | lo]
        hi = v.to(torch.float16)
        lo = (v - hi.float()).to(torch.float16)
        mups = torch.cat([hi, lo], dim=-1).contiguous()
    net.gate(mups)
    torch.cuda.synchronize()
    lib.nesti_profile_enable(1)
    for _ in range(reps):
        net.gate(mups)
    torch.cuda.synchronize()
    ms, _ = _lib.profile_read(lib)
    lib.nesti_profile_enable(0)
    out = {c: sum(ms[ph][c] for ph in _lib.PROF_PHASES) / reps for c in _lib.PROF_CONV}
    out["total"] = sum(out.values())
    print("AB_RESULT " + json.dumps(out))


def main():
    if sys.argv[1] == "--worker":
        return worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    B, reps, cfgs = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3:]
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    log = open(os.path.join(REPO, "gpurun_out", "ab_gate.txt"), "a")
    for rnd in (1, 2):
        for c in cfgs:
            envs, dtype = c.rsplit(":", 1)
            env = dict(os.environ)
            for kv in filter(None, envs.split(",")):
                k, v = kv.split("=", 1)
                env[k] = v
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(B), str(reps), dtype], env=env,
                               capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("AB_RESULT ")]
            if r.returncode != 0 or not line:
                msg = "%-44s FAILED rc=%d %s" % (c, r.returncode, (r.stderr or r.stdout)[-400:])
            else:
                d = json.loads(line[-1][len("AB_RESULT "):])
                msg = "%-44s r%d B=%d  total %8.2f ms | " % (c, rnd, B, d["total"]) + "  ".join("%s %7.2f" % (k, d[k]) for k in d if k != "total")
            print(msg, flush=True)
            log.write(msg + "\n")
            log.flush()


if __name__ == "__main__":
    main()
