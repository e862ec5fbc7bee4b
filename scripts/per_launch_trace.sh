#!/bin/bash
# One pass of the default bench workload under rocprofv3 --kernel-trace; prints every conv / pool / MuPS dispatch in
# launch order (gate tower first, then the experts 0..6) with its grid and duration -> gpurun_out/per_launch.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
DT=${1:-f16x3c}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/plt && rocprofv3 --kernel-trace --output-format csv -d /tmp/plt -- python3 $R/bench.py --streams 1 --dtype $DT --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-secondary --no-kernel-timing > /tmp/plt.json 2> /tmp/plt.err
f=$(find /tmp/plt -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/per_launch_$DT.txt <<'P'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = [r for r in rows if re.search(r"conv8n_kernel|conv4n_kernel|conv_igemm_kernel|maxpool|mups", r["Kernel_Name"])]
# the timed pass is the LAST 134 conv launches (+ pools); print everything after the last mups launch
last_mups = max(i for i, r in enumerate(keep) if "mups" in r["Kernel_Name"])
tot = {}
for i, r in enumerate(keep[last_mups:]):
    name = re.sub(r"void nesti::\(anonymous namespace\)::|\(nesti::.*", "", r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
    tot[name] = tot.get(name, 0.0) + d
    print("%3d %-40s wgs %8d  %9.3f ms" % (i, name, g, d))
print(tot)
P
tail -3 $O/per_launch_$DT.txt | cut -c1-400
