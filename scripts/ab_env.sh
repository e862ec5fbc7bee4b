#!/bin/bash
# same-box A/B over environment settings: scripts/ab_env.sh <outdir> <dtype> "VAR=a" "VAR=b" ...
out=gpurun_out/$1; dt=$2; shift 2; mkdir -p $out
for rep in 1 2; do
  i=0
  for setting in "$@"; do
    i=$((i+1))
    env $setting python bench.py --steps 2 --warmup 1 --dtype $dt --no-parity --no-secondary --no-cpu-baseline > $out/ab_${i}_r${rep}.json 2> $out/ab_${i}_r${rep}.err
    python - <<PY
import json
d=json.loads(open("$out/ab_${i}_r${rep}.json").read().strip().splitlines()[-1])
print("%-40s rep $rep $dt: %.0f normals/s, conv %.1f ms/step, frac %.4f" % ("$setting", d["value"], d["roofline"]["kernel_ms_per_step"]["conv"], d["roofline"]["frac"]))
PY
  done
done
