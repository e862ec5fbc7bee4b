#!/bin/bash
# SQ counters of the gating tower's conv launches (last repetition) at batch $1; extra env as $2 (e.g. NESTI_LIB=<another build>, or "" );
# dtype as $3 (f16 / f16x3 / bf16 ...); PROF_DRIVER=prof_expert.py profiles the expert towers instead (scripts/prof_expert.py)
cd /tmp && export TMPDIR=/tmp
B=${1:-2048}
DT=${3:-f16}
DRV=${PROF_DRIVER:-prof_gate.py}
[ -n "$2" ] && export $2
out=/tmp/pg_$RANDOM
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/scripts/$DRV $B 2 $DT > $out.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d ${out}b -- python3 $GRAFT_REPO_ROOT/scripts/$DRV $B 2 $DT > ${out}b.log 2>&1
python3 - $out ${out}b <<'PY'
import csv, sys, glob, collections
def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    by = collections.OrderedDict()
    for r in rows:
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"].replace("void nesti::(anonymous namespace)::", "").split("(")[0], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        by.setdefault(k, {})[r["Counter_Name"]] = by.get(k, {}).get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    return by
a, b = load(sys.argv[1]), load(sys.argv[2])
ka = [k for k in a if "conv" in k[1]]; kb = [k for k in b if "conv" in k[1]]
n = len(ka) // 2
for k, k2 in list(zip(ka, kb))[n:n + int(__import__('os').environ.get('PROF_ROWS', '18'))]:
    c, d = a[k], b[k2]
    wc = c["SQ_WAVE_CYCLES"]
    cyc = 16.0 if "conv4n" in k[1] else 32.0                                      # cycles per MFMA: 32x32x16 (32) or conv4n_kernel's 16x16x32 (16)
    if "conv8n_kernel<2, 5, 2" in k[1]: cyc = (40 * 32.0 + 24 * 64.0) / 64          # the FP8 cross-term loop: per row of taps 40 f16 MFMAs (32 cycles) + 24 FP8 K=64 (64)
    if "conv8n_kernel<2, 3, 2" in k[1]: cyc = (24 * 32.0 + 16 * 64.0) / 40
    busy = c["SQ_INSTS_MFMA"] * cyc / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)      # 1024 SIMDs, GUI_ACTIVE summed over 8 XCDs
    print("%-34s wgs %5d | mfma %.3g (pipe busy %4.1f %%) valu %.3g salu %.3g lds %.3g | wave time: wait %.2f issue-stall %.2f active %.2f | lds conflict %.3f" % (
        k[1], k[2], c["SQ_INSTS_MFMA"], 100 * busy, c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"], c["SQ_INSTS_LDS"], c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc,
        d["SQ_LDS_BANK_CONFLICT"] / max(1, d["SQ_LDS_IDX_ACTIVE"])))
PY
