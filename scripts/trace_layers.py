"""Print the per-launch durations of the LAST forward pass found in a rocprofv3 kernel-trace CSV
(scripts/prof_forward.py run under `rocprofv3 --kernel-trace --output-format csv`)."""
import csv
import sys


def last_forward(path):
    rows = list(csv.DictReader(open(path)))
    ks = [(r["Kernel_Name"].replace("void nesti::(anonymous namespace)::", "").split("(")[0],
           (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
           int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) for r in rows]
    start = [i for i, k in enumerate(ks) if "mups" in k[0]][-1]
    return ks[start:]


if __name__ == "__main__":
    runs = [last_forward(p) for p in sys.argv[1:]]
    for i, k in enumerate(runs[0]):
        if k[1] < 40 and all(r[i][1] < 40 for r in runs if i < len(r)):
            continue
        print("%3d %-40s wgs %6d " % (i, k[0], k[2]) + "  ".join("%9.1f us" % r[i][1] for r in runs if i < len(r)))
    print("total ms: " + "  ".join("%.2f" % (sum(k[1] for k in r) / 1e3) for r in runs))
