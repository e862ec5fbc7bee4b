"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, csv) into
profiles/<round>_pmc_traffic.json.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section), so the read
side is doubled.  Usage: summarize_pmc.py <dir with pmc_FETCH_SIZE/, pmc_WRITE_SIZE/> <queries> <out.json>"""
import collections
import csv
import json
import sys

root, queries, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]


def load(c):
    d, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open("%s/pmc_%s/p_counter_collection.csv" % (root, c))):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"]
        key = "conv_igemm_kernel" if "conv_igemm" in k else "mups_kernel" if "mups" in k else \
            "maxpool2_kernel" if "maxpool" in k else "patches_kernel" if "patches_kernel" in k else "other"
        d[key] += float(r["Counter_Value"])
        n[key] += 1
    return d, n


f, nf = load("FETCH_SIZE")
w, nw = load("WRITE_SIZE")
res = {"queries": queries, "note": "FETCH_SIZE doubled per the gfx950 correction; units KiB -> bytes", "kernels": {}}
for k in sorted(set(f) | set(w)):
    rd, wr = 2.0 * f[k] * 1024, w[k] * 1024
    res["kernels"][k] = {"launches": nf[k], "hbm_read_bytes": rd, "hbm_write_bytes": wr,
                         "hbm_bytes_per_launch": (rd + wr) / max(1, nf[k]), "hbm_bytes_per_query": (rd + wr) / queries}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["kernels"]["conv_igemm_kernel"]))
