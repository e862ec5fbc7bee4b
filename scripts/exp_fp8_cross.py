"""VERDICT r05 item 1, step 0 -- numerics before any kernel: may the two cross terms of the experts' three-product scheme go
through e4m3?

Today every multiply of an expert tower is  hi*W_hi + lo*W_hi + hi*W_lo  (f16 pairs, fp32 accumulate: conv8n.hip tile_mma).
The proposal keeps hi*W_hi in f16 and computes BOTH cross terms in one block-scaled FP8 MFMA
(v_mfma_scale_f32_32x32x64_f8f6f4, same fp32 accumulator):  e4m3(lo 2^sa) * e4m3(W_hi 2^sb) + e4m3(hi 2^sc) * e4m3(W_lo 2^sd),
power-of-two scales per layer with sa + sb = sc + sd (one E8M0 block scale per operand).

This script is an EMULATION in torch on the GPU (no new kernel): the expert towers of models/experts_n_est.py:243-291 with
every layer's arithmetic spelled out the way the library does it -- BN folded (utils/tf_util.py:491-494), weights scaled by the
packer's power of two (model.hip pack_layer), activations and weights as f16 pairs, products exact, fp32 accumulation, bias +
ReLU in fp32, outputs split into pairs again -- and, for the layers of a variant, the cross terms rounded to e4m3
(torch.float8_e4m3fn) or, for information, to a block-scaled e2m3 (FP6) with one scale per 16 channels.
For every query of the bench's 100k cloud (calibrated gate, the library's f16x3 gate decisions) the routed expert is evaluated
  * by the LIBRARY in f16x3 (anchors the emulation: the emulated three-product tower must agree with it to ~1e-7),
  * by the emulated three-product tower (the reference of this experiment),
  * by the emulated single-product variant of inception2's 5^3 layer (anchors the error model against
    profiles/r05_expert_mix.txt: p50 8.5e-8, p99 3.2e-6, max 8.6e-4 measured with the real kernels),
  * by each FP8 variant.
Printed per variant: 1 - cos (p50 / p99 / p99.9 / max) and |dn| against the emulated three-product tower.
GATE: max 1 - cos <= 2.5e-6 on the 100k cloud.   Writes gpurun_out/fp8_cross.{json,txt} (-> profiles/r06_fp8_cross.txt).

    python scripts/exp_fp8_cross.py                # GPU box
    FP8X_POINTS=64 python scripts/exp_fp8_cross.py --cpu    # plumbing check without a GPU (random MuPS, random routing)
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nesti_net_amd  # noqa: E402,F401
from nesti_net_amd import synth, weights  # noqa: E402
from nesti_net_amd.config import NestiConfig  # noqa: E402

CPU = "--cpu" in sys.argv
dev = torch.device("cpu" if CPU else "cuda:0")
BN_EPS = 1e-3
E4M3_MAX = 448.0

# ---- the variants: which tap layers (by scope suffix) take their cross terms through 8 bits --------------------------------
I1C2, I1C3, I2C2, I2C3 = "inception1%s_conv2", "inception1%s_conv3", "inception2%s_conv2", "inception2%s_conv3"
I4C2, I4C3, I6C2, I6C3 = "inception4%s_conv2", "inception4%s_conv3", "inception6%s_conv2", "inception6%s_conv3"
VARIANTS = [
    ("single i2 conv3", "single", [I2C3]),                       # anchor: ONE product in that layer (r05_expert_mix mask 001000)
    ("e4m3 i2 conv3", "e4m3", [I2C3]),
    ("e4m3 both 5^3", "e4m3", [I1C3, I2C3]),
    ("e4m3 all taps at 8^3", "e4m3", [I1C2, I1C3, I2C2, I2C3]),
    ("e4m3 all taps", "e4m3", [I1C2, I1C3, I2C2, I2C3, I4C2, I4C3, I6C2, I6C3]),
    ("e4m3 all taps, lo only", "e4m3_lo", [I1C2, I1C3, I2C2, I2C3, I4C2, I4C3, I6C2, I6C3]),   # hi*W_lo stays f16: which cross term carries the error?
    ("e2m3 blocks all taps at 8^3", "e2m3", [I1C2, I1C3, I2C2, I2C3]),
    # the form the instruction can actually run on this kernel's rows: a lane's K block is one tap's [lo 16 ch | hi 16 ch] with ONE scale,
    # taken from max |hi| of the 16 channels (|lo 2^11| <= |hi| element by element), weights likewise per (tap, 16 ch, column)
    ("e2m3 shared-scale all taps at 8^3", "e2m3s", [I1C2, I1C3, I2C2, I2C3]),
    ("e2m3 shared-scale both 5^3", "e2m3s", [I1C3, I2C3]),
]
if os.environ.get("FP8X_VARIANTS"):
    VARIANTS = [v for v in VARIANTS if any(t in v[0] for t in os.environ["FP8X_VARIANTS"].split(","))]


def q_e4m3(x):
    return x.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def q_e2m3_blocks(x, block=16):
    """Block-scaled FP6 e2m3 along the last axis: one power-of-two scale per `block` elements (the OPTIMISTIC form: lo and hi
    blocks scaled separately), grid {0, .125 .. 7.5}."""
    shp = x.shape
    pad = (-shp[-1]) % block                                            # the library pads channels with zeros
    if pad:
        x = torch.nn.functional.pad(x, (0, pad))
    v = x.reshape(-1, block)
    amax = v.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    e = torch.ceil(torch.log2(amax / 7.5))                              # amax / 2^e <= 7.5
    s = torch.exp2(e)
    m = (v / s).abs()
    be = torch.floor(torch.log2(m.clamp_min(1e-30))).clamp(0, 2)        # binade of the element: [1,2) [2,4) [4,8); below 1: step of [1,2)
    step = torch.exp2(be - 3)
    qv = (torch.round(m / step) * step).clamp(max=7.5) * torch.sign(v) * s
    qv = qv.reshape(shp[:-1] + (shp[-1] + pad,))
    return qv[..., :shp[-1]].contiguous() if pad else qv


def e2m3_round(m):
    """Magnitudes (already divided by the block scale) to the e2m3 grid {0, .125 .. 7.5}, saturating."""
    be = torch.floor(torch.log2(m.clamp_min(1e-30))).clamp(0, 2)
    step = torch.exp2(be - 3)
    return (torch.round(m / step) * step).clamp(max=7.5)


def q_e2m3_shared(hi, lo, block=16):
    """[hi | lo 2^11] of `block` channels under ONE power-of-two scale 2^(E - 2), E = exponent of the block's largest |hi| (so the largest
    element lands in [4, 8) and saturates at 7.5).  Returns the dequantised (hi6, lo6) in true units."""
    shp = hi.shape
    pad = (-shp[-1]) % block
    if pad:
        hi, lo = torch.nn.functional.pad(hi, (0, pad)), torch.nn.functional.pad(lo, (0, pad))
    h, l = hi.reshape(-1, block), lo.reshape(-1, block) * 2048.0
    amax = h.abs().amax(dim=1, keepdim=True)
    s = torch.exp2(torch.floor(torch.log2(amax.clamp_min(1e-30))) - 2)
    hq = e2m3_round(h.abs() / s) * torch.sign(h) * s
    lq = e2m3_round(l.abs() / s) * torch.sign(l) * s / 2048.0
    hq, lq = hq.reshape(shp[:-1] + (shp[-1] + pad,)), lq.reshape(shp[:-1] + (shp[-1] + pad,))
    return (hq[..., :shp[-1]].contiguous(), lq[..., :shp[-1]].contiguous()) if pad else (hq, lq)


def split16(x):
    hi = x.to(torch.float16).to(torch.float32)
    lo = (x - hi).to(torch.float16).to(torch.float32)
    return hi, lo


def pow2_floor_exp(v):
    return int(np.floor(np.log2(v))) if v > 0 else 0


class Layer:
    """One conv / fc layer with BN folded and the packer's power-of-two weight scale."""

    def __init__(self, W, scope, bn=True, relu=True):
        w = np.asarray(W[scope + "/weights"], np.float64)
        b = np.asarray(W[scope + "/biases"], np.float64)
        if bn:
            inv = np.asarray(W[scope + "/bn/gamma"], np.float64) / np.sqrt(np.asarray(W[scope + "/bn/var"], np.float64) + BN_EPS)
            b = (b - np.asarray(W[scope + "/bn/mean"], np.float64)) * inv + np.asarray(W[scope + "/bn/beta"], np.float64)
            w = (w * inv).astype(np.float32)               # the packer multiplies in fp32: wrow[n] * scale[n] * wmul
        w = w.astype(np.float32)
        wmax = float(np.abs(w).max())
        m, e = np.frexp(wmax)
        e = int(min(24, max(-8, 14 - e)))                  # model.hip pack_layer: wmax 2^e in [2^13, 2^14)
        self.acc_scale = float(2.0 ** -e)
        ws = torch.as_tensor(w * np.float32(2.0 ** e), device=dev)
        self.k = w.shape[0] if w.ndim == 5 else 1
        ws = ws.reshape(-1, w.shape[-2], w.shape[-1])      # [taps, cin, cout]
        self.w_hi, self.w_lo = split16(ws)
        # e4m3 images: W_hi 2^sb up to 256 (< 448), W_lo (<= half an ulp of 2^14 = 2^3) 2^sd up to 256
        self.sb, self.sd = -6, 5
        self.w_hi8 = q_e4m3(self.w_hi * 2.0 ** self.sb)
        self.w_lo8 = q_e4m3(self.w_lo * 2.0 ** self.sd)
        # FP6: blocks of 16 input channels per (tap, output column)
        self.w_hi6 = q_e2m3_blocks(self.w_hi.transpose(1, 2).contiguous()).transpose(1, 2).contiguous()
        self.w_lo6 = q_e2m3_blocks(self.w_lo.transpose(1, 2).contiguous()).transpose(1, 2).contiguous()
        h6, l6 = q_e2m3_shared(self.w_hi.transpose(1, 2).contiguous(), self.w_lo.transpose(1, 2).contiguous())
        self.w_hi6s, self.w_lo6s = h6.transpose(1, 2).contiguous(), l6.transpose(1, 2).contiguous()
        self.bias = torch.as_tensor(b.astype(np.float32), device=dev)
        self.relu = relu
        self.scope = scope
        self.amax = 0.0                                    # largest input activation seen (calibration batch)
        self.sc = None                                     # e4m3 scale exponents of the input planes, fixed after calibration


def taps_matmul(x, w, k):
    """sum over the k^3 taps of [rows, Cin] @ [Cin, Cout] on the zero-padded volume (TF SAME: floor((k-1)/2) on the low side).
    x [B,D,D,D,C] fp32, w [k^3, Cin, Cout] -> [B*D^3, Cout] fp32 (fp32 FMA accumulation, like the MFMA's accumulator)."""
    B, D = x.shape[0], x.shape[1]
    C = x.shape[-1]
    if k == 1:
        return x.reshape(-1, C) @ w[0]
    lo = (k - 1) // 2
    hi = k - 1 - lo
    xp = torch.nn.functional.pad(x, (0, 0, lo, hi, lo, hi, lo, hi))
    out = torch.zeros((B * D * D * D, w.shape[2]), dtype=x.dtype, device=x.device)
    t = 0
    for a in range(k):
        for b in range(k):
            for c in range(k):
                if abs(a - lo) < D and abs(b - lo) < D and abs(c - lo) < D:
                    out.addmm_(xp[:, a:a + D, b:b + D, c:c + D, :].reshape(-1, C), w[t])
                t += 1
    return out


def run_layer(L, x, mode, calibrate=False):
    """x = (hi, lo) [B,D,D,D,C] (or [B,C] for fc) -> activated fp32 output of the layer, same leading shape."""
    hi, lo = x
    fc = hi.dim() == 2
    if fc:
        hi, lo = hi[:, None, None, None, :], lo[:, None, None, None, :]
    if calibrate:
        L.amax = max(L.amax, float(hi.abs().max()))
    k = L.k
    if mode == "x3":
        acc = taps_matmul(hi + lo, L.w_hi, k) + taps_matmul(hi, L.w_lo, k)
    elif mode == "single":
        acc = taps_matmul(hi, L.w_hi, k)
    else:
        if L.sc is None:
            raise RuntimeError("layer %s has no calibrated activation scale" % L.scope)
        acc = taps_matmul(hi, L.w_hi, k)
        sc = L.sc                                       # hi 2^sc <= 256; lo (<= 2^-11 hi) 2^(sc + 11)
        sa = sc + 11
        if mode in ("e4m3", "e4m3_lo"):
            acc = acc + taps_matmul(q_e4m3(lo * 2.0 ** sa), L.w_hi8, k) * 2.0 ** -(sa + L.sb)
            if mode == "e4m3":
                acc = acc + taps_matmul(q_e4m3(hi * 2.0 ** sc), L.w_lo8, k) * 2.0 ** -(sc + L.sd)
            else:
                acc = acc + taps_matmul(hi, L.w_lo, k)
        elif mode == "e2m3":
            acc = acc + taps_matmul(q_e2m3_blocks(lo), L.w_hi6, k) + taps_matmul(q_e2m3_blocks(hi), L.w_lo6, k)
        elif mode == "e2m3s":
            h6, l6 = q_e2m3_shared(hi, lo)
            acc = acc + taps_matmul(l6, L.w_hi6s, k) + taps_matmul(h6, L.w_lo6s, k)
        else:
            raise ValueError(mode)
    y = acc * L.acc_scale + L.bias
    if L.relu:
        y = torch.relu(y)
    shp = (hi.shape[0],) if fc else tuple(hi.shape[:4])
    return y.reshape(shp + (y.shape[-1],))


def avg_pool_same(x, k):
    """tf.nn.avg_pool3d k^3 stride 1 SAME: mean over the taps inside the volume (utils/tf_util.py:450-454)."""
    if k == 1:
        return x
    lo = (k - 1) // 2
    hi = k - 1 - lo
    xc = torch.nn.functional.pad(x.permute(0, 4, 1, 2, 3), (lo, hi, lo, hi, lo, hi))
    s = torch.nn.functional.avg_pool3d(xc, k, stride=1) * float(k ** 3)
    ones = torch.nn.functional.pad(torch.ones((1, 1) + tuple(x.shape[1:4]), dtype=x.dtype, device=x.device), (lo, hi, lo, hi, lo, hi))
    cnt = torch.nn.functional.avg_pool3d(ones, k, stride=1) * float(k ** 3)
    return (s / cnt).permute(0, 2, 3, 4, 1)


def max_pool2(x):
    return torch.nn.functional.max_pool3d(x.permute(0, 4, 1, 2, 3), 2, 2).permute(0, 2, 3, 4, 1)


class Expert:
    def __init__(self, W, i):
        s = "Expert_%d" % i
        self.s = s
        self.blocks = []
        for name, k0, k1 in (("inception1", 3, 5), ("inception2", 3, 5), ("inception4", 2, 4), ("inception6", 2, 4)):
            sc = name + s
            self.blocks.append((k0, [Layer(W, sc + "_conv%d" % j) for j in (1, 2, 3, 4)]))
        self.fcs = [Layer(W, "fc1" + s), Layer(W, "fc2" + s), Layer(W, "fc3" + s), Layer(W, "fc4" + s, bn=False, relu=False)]

    def layers(self):
        for _, ls in self.blocks:
            yield from ls
        yield from self.fcs

    def finish_calibration(self):
        for L in self.layers():
            # hi 2^sc in (128, 256] for the largest activation of the calibration batch: one binade of headroom below 448
            L.sc = 8 - int(np.ceil(np.log2(max(L.amax, 1e-20))))

    def forward(self, mups, modes, calibrate=False):
        """mups [B,8,8,8,C] fp32 (this expert's scales) -> [B,3].  modes: {scope: mode} for the layers that are not 'x3'."""
        def run(L, x):
            return run_layer(L, x, modes.get(L.scope, "x3"), calibrate)
        x = split16(mups)
        for bi, (k0, (c1, c2, c3, c4)) in enumerate(self.blocks):
            y1 = run(c1, x)
            p1 = split16(y1)
            y2 = run(c2, p1)
            y3 = run(c3, p1)
            # conv4 = 1x1x1 conv of the avg-pooled input; the library averages the fp32 accumulators of the 1x1x1 conv instead
            # (avg-pool commutes with it: conv.hip) -- emulated in that order
            hi, lo = x
            acc4 = taps_matmul(hi + lo, c4.w_hi, 1) + taps_matmul(hi, c4.w_lo, 1)
            acc4 = avg_pool_same(acc4.reshape(tuple(hi.shape[:4]) + (-1,)), k0)
            y4 = torch.relu(acc4 * c4.acc_scale + c4.bias)
            y = torch.cat([y1, y2, y3, y4], dim=4)
            if bi >= 1:
                y = max_pool2(y)
            x = split16(y)
        g = (x[0].reshape(x[0].shape[0], -1), x[1].reshape(x[1].shape[0], -1))
        for L in self.fcs[:-1]:
            g = split16(run(L, g))
        return run(self.fcs[-1], g)


def main():
    torch.backends.cuda.matmul.allow_tf32 = False
    cfg = NestiConfig()
    N = int(os.environ.get("FP8X_POINTS", "100000"))
    B = int(os.environ.get("FP8X_BATCH", "4096"))
    E = cfg.n_experts
    W = weights.synthetic_weights(cfg)
    t_start = time.time()
    if CPU:
        rng = np.random.RandomState(0)
        mups_all = torch.as_tensor(np.abs(rng.randn(N, 8, 8, 8, 60)).astype(np.float32) * 0.05)
        expert_all = rng.randint(0, E, size=N)
        lib_out = None
        net = cp = None
    else:
        from nesti_net_amd.calibrate import calibrate_gate
        from nesti_net_amd.model import NestiNet, mups_forward
        from nesti_net_amd.provider import CloudPatches
        pts = synth.make_cloud("ellipsoid", n=N, seed=1234)[0]
        cp = CloudPatches(pts, cfg, device=dev)
        sp, sn = cp.build(0, 512)
        W = calibrate_gate(cfg, W, sp, sn, device=dev)
        del sp, sn
        net = NestiNet(cfg, W, dtype="f16x3", device=dev, max_batch=B)
    experts = [Expert(W, i) for i in range(E)]
    scopes = {i: {v[0]: {t % experts[i].s: v[1] for t in v[2]} for v in VARIANTS} for i in range(E)}

    names = ["x3"] + [v[0] for v in VARIANTS]
    outs = {n: [] for n in names}
    lib_all, expert_rows = [], []
    calibrated = False
    for done in range(0, N, B):
        take = min(B, N - done)
        if CPU:
            mups = mups_all[done:done + take]
            expert = torch.as_tensor(expert_all[done:done + take])
        else:
            p, n = cp.build(done, take)
            mups_l = net.mups(p, n)
            _, expert = net.gate(mups_l)
            lib_all.append(net.experts(mups_l, expert).double().cpu().numpy())
            mups = mups_forward(cfg, p, n, out_dtype="f32")
            del mups_l, p, n
        if not calibrated:                                   # activation ranges from the first batch (the product would calibrate
            for i in range(E):                               # them like the gate margin: once per model on a sample of queries)
                rows = torch.nonzero(expert == i).flatten()
                if len(rows):
                    lo_c = min(cfg.expert_dict[i]) * 20
                    experts[i].forward(mups[rows][..., lo_c:lo_c + 20 * len(cfg.expert_dict[i])], {}, calibrate=True)
            for ex in experts:
                ex.finish_calibration()
            calibrated = True
        res = {n: torch.zeros((take, 3), dtype=torch.float32, device=dev) for n in names}
        for i in range(E):
            rows = torch.nonzero(expert == i).flatten()
            if not len(rows):
                continue
            lo_c = min(cfg.expert_dict[i]) * 20
            m_i = mups[rows][..., lo_c:lo_c + 20 * len(cfg.expert_dict[i])].contiguous()
            res["x3"][rows] = experts[i].forward(m_i, {})
            for v in VARIANTS:
                res[v[0]][rows] = experts[i].forward(m_i, scopes[i][v[0]])
        for n in names:
            outs[n].append(res[n].double().cpu().numpy())
        expert_rows.append(expert.cpu().numpy())
        print("  %d / %d queries, %.0f s" % (done + take, N, time.time() - t_start), flush=True)

    ref = np.concatenate(outs["x3"])
    nref = np.linalg.norm(ref, axis=1)
    lines = []
    result = {"queries": N, "routing": np.bincount(np.concatenate(expert_rows), minlength=E).tolist(), "variants": [],
              "activation_scale_exponents": {L.scope: L.sc for ex in experts for L in ex.layers() if L.k > 1}}

    def q(v, x):
        return float(np.quantile(v, x))

    def stats(a, b):
        na, nb = np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1)
        omc = 1.0 - (a * b).sum(1) / np.maximum(na * nb, 1e-300)
        dn = np.linalg.norm(a - b, axis=1)
        return omc, dn

    lines.append("FP8 cross terms, step 0 (scripts/exp_fp8_cross.py): %d queries of the bench cloud, routing %s; |n| p1 %.3g p50 %.3g min %.3g"
                 % (N, result["routing"], q(nref, .01), q(nref, .5), nref.min()))
    if lib_all:
        lib = np.concatenate(lib_all)
        omc, dn = stats(lib, ref)
        result["emulated_x3_vs_library_f16x3"] = {"one_minus_cos_max": float(omc.max()), "one_minus_cos_p99": q(omc, .99), "dn_max": float(dn.max())}
        lines.append("anchor 1: emulated three-product tower vs the LIBRARY's f16x3 experts: 1-cos p99 %.3g max %.3g, |dn| max %.3g"
                     % (q(omc, .99), omc.max(), dn.max()))
    for v in VARIANTS:
        out = np.concatenate(outs[v[0]])
        omc, dn = stats(out, ref)
        ent = {"variant": v[0], "mode": v[1], "layers": [t % "<E>" for t in v[2]],
               "one_minus_cos": {"p50": q(omc, .5), "p99": q(omc, .99), "p999": q(omc, .999), "max": float(omc.max())},
               "dn": {"p50": q(dn, .5), "p99": q(dn, .99), "max": float(dn.max())},
               "frac_over_2.5e-6": float((omc > 2.5e-6).mean()), "frac_over_1e-5": float((omc > 1e-5).mean()),
               "passes_2.5e-6": bool(omc.max() <= 2.5e-6)}
        result["variants"].append(ent)
        lines.append("%-30s 1-cos p50 %.3g p99 %.3g p99.9 %.3g max %.3g   |dn| p50 %.3g p99 %.3g max %.3g   over 2.5e-6: %d queries  %s"
                     % (v[0], ent["one_minus_cos"]["p50"], ent["one_minus_cos"]["p99"], ent["one_minus_cos"]["p999"], ent["one_minus_cos"]["max"],
                        ent["dn"]["p50"], ent["dn"]["p99"], ent["dn"]["max"], int((omc > 2.5e-6).sum()),
                        "" if v[1] == "single" else ("PASS" if ent["passes_2.5e-6"] else "FAIL")))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(result, open("gpurun_out/fp8_cross.json", "w"), indent=1)
    open("gpurun_out/fp8_cross.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    with torch.no_grad():
        main()
