#!/bin/bash
# SQ counters of mups_kernel under bench.py --mups-only (BASELINE config 1)
cd /tmp && export TMPDIR=/tmp
out=/tmp/pm_$RANDOM
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --mups-only --steps 1 --warmup 1 > $out.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d ${out}b -- python3 $GRAFT_REPO_ROOT/bench.py --mups-only --steps 1 --warmup 1 > ${out}b.log 2>&1
python3 - $out ${out}b <<'PY'
import csv, sys, glob, collections
def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(float); n = 0
    for r in csv.DictReader(open(f)):
        if "mups_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    return agg
a, b = load(sys.argv[1]), load(sys.argv[2])
wc = a["SQ_WAVE_CYCLES"]
print("mups_kernel (all launches): VALU insts %.4g  LDS insts %.4g  SALU %.4g" % (a["SQ_INSTS_VALU"], a["SQ_INSTS_LDS"], a["SQ_INSTS_SALU"]))
print("wave time: wait %.2f  issue-stall %.2f  active %.2f ; VALU-active share of wave cycles %.2f" % (a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc, a["SQ_ACTIVE_INST_VALU"] / wc))
gui = b["GRBM_GUI_ACTIVE"] / 8.0
print("per-SIMD VALU instruction rate: %.3f wave-instructions per cycle (GUI cycles per XCD %.4g, 1024 SIMDs)" % (a["SQ_INSTS_VALU"] / (gui * 1024.0), gui))
print("LDS: bank-conflict share %.3f, LDS-issue-stall share of wave cycles %.3f" % (b["SQ_LDS_BANK_CONFLICT"] / max(1.0, b["SQ_LDS_IDX_ACTIVE"]), b["SQ_WAIT_INST_LDS"] / wc))
PY
