#!/bin/bash
# per-launch durations of the gating tower (last repetition) at batch $1 under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp
B=${1:-8192}; [ -n "$2" ] && export $2
rm -rf /tmp/gl && rocprofv3 --kernel-trace --output-format csv -d /tmp/gl -- python3 $GRAFT_REPO_ROOT/scripts/prof_gate.py $B 2 > /tmp/gl.log 2>&1
f=$(find /tmp/gl -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$B" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
B = int(sys.argv[2])
ks = [(r["Kernel_Name"].replace("void nesti::(anonymous namespace)::", "").split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) for r in rows]
n = len(ks) // 2
last = ks[-n:]
tot = 0
for i, k in enumerate(last):
    tot += k[1]
    print("%3d %-36s wgs %6d %10.1f us  %7.3f us/pt" % (i, k[0], k[2], k[1], k[1] / B))
print("total %.2f ms, %.3f us/pt" % (tot / 1e3, tot / B))
PY
