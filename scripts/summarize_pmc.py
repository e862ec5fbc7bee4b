"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, csv) of ONE bench step into
profiles/<round>_pmc_traffic.json.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section), so the read side is doubled.  Only the timed
dtype's launches are counted (the gate-calibration probe runs a few f32 launches, template argument 0).
Usage: summarize_pmc.py <dir with pmc_FETCH_SIZE/, pmc_WRITE_SIZE/> <queries> <out.json> <dtype> <batch>"""
import collections
import csv
import json
import re
import sys

root, queries, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
dtype = sys.argv[4] if len(sys.argv) > 4 else "f16"
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 25000
DT = {"f32": "0", "bf16": "1", "f16": "2", "f16x3": "2", "f16x3c": "2", "f16x8": "2", "f16x8c": "2", "bf16x3": "1"}[dtype]     # the kernels' element type


def family(k):
    # the fused product-path kernel first: its template argument is the OUTPUT dtype (4 = the f16 pair layout), not a conv
    # element type, and its name contains "mups_kernel" (the row used to be dropped by the filter below)
    if "patches_mups_kernel" in k:
        return "patches_mups_kernel"
    m = re.search(r"(conv8n_kernel|conv4n_kernel|conv_igemm_kernel|mups_kernel|maxpool2_kernel)<(\d)", k)
    if m:
        if m.group(2) != DT:
            return None
        return "conv" if m.group(1).startswith("conv") else m.group(1)
    if "patches_kernel" in k:
        return "patches_kernel"
    return "other"


def klass(k):
    """conv launches by kernel class and K loop: conv8 5^3 / 3^3, taps (igemm, several taps), 1x1 + FC (igemm, one tap)."""
    m = re.search(r"conv8n_kernel<(\d), (\d), (\d|true|false)", k)        # third argument: 0 plain, 1 the pair K loop, 2 the FP8 cross-term loop
    if m:
        return "conv8_k%s%s" % (m.group(2), {"true": "_pair", "1": "_pair", "2": "_x8", "3": "_x8"}.get(m.group(3), ""))
    m = re.search(r"conv4n_kernel<(\d), (\d), (true|false)", k)
    if m:
        return "taps_4_2" + ("_pair" if m.group(3) == "true" else "")
    m = re.search(r"conv_igemm_kernel<(\d), \d+, (true|false), (true|false)", k)
    if m:
        return ("one_by_one_fc" if m.group(2) == "true" else "taps_4_2") + ("_pair" if m.group(3) == "true" else "")
    return None


def load(c):
    d, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open("%s/pmc_%s/p_counter_collection.csv" % (root, c))):
        if r["Counter_Name"] != c:
            continue
        key = family(r["Kernel_Name"])
        if key is None:
            continue
        d[key] += float(r["Counter_Value"])
        n[key] += 1
        if key == "conv" and klass(r["Kernel_Name"]):
            d["conv:" + klass(r["Kernel_Name"])] += float(r["Counter_Value"])
            n["conv:" + klass(r["Kernel_Name"])] += 1
    return d, n


f, nf = load("FETCH_SIZE")
w, nw = load("WRITE_SIZE")
res = {"queries": queries, "dtype": dtype, "batch": batch, "calibrated_gate": True,
       "command": "bench.py --streams 1 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-parity --no-secondary under rocprofv3 --pmc (one pass over the cloud)",
       "note": "FETCH_SIZE doubled per the gfx950 correction; units KiB -> bytes; 'conv' = conv8n_kernel + conv4n_kernel + conv_igemm_kernel (f16x3c: includes the gate-margin calibration's 1024-query double pass, ~1 %)", "kernels": {}}
for k in sorted(set(f) | set(w)):
    rd, wr = 2.0 * f[k] * 1024, w[k] * 1024
    res["kernels"][k] = {"launches": nf[k], "hbm_read_bytes": rd, "hbm_write_bytes": wr,
                         "hbm_bytes_per_launch": (rd + wr) / max(1, nf[k]), "hbm_bytes_per_query": (rd + wr) / queries}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["kernels"].get("conv")))
