#!/bin/bash
# per-launch durations of the gate tower (last repetition), for the library in $1 and dtype $2
cd /tmp && export TMPDIR=/tmp
export NESTI_LIB=$1
rm -rf /tmp/tg && rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -- python3 $GRAFT_REPO_ROOT/scripts/prof_gate.py 32768 2 $2 > /tmp/tg.log 2>&1
f=$(find /tmp/tg -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = [r for r in rows if re.search(r"conv_igemm_kernel", r["Kernel_Name"])]
n = len(keep) // 2
for r in keep[n:]:
    name = re.sub(r"void nesti::\(anonymous namespace\)::|\(nesti::.*", "", r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
    print("%-46s wgs %7d %8.3f ms" % (name, g, d))
P
