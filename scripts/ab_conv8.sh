#!/bin/bash
# same-box A/B of the conv kernels: NESTI_CONV8 = 0 (conv_igemm_kernel only) / 1 (5^3 layers on conv8_kernel) / 2 (3^3 too)
out=gpurun_out/${1:-ab8}; mkdir -p $out
dt=${2:-bf16}
for rep in 1 2; do
  for m in 0 1 2; do
    NESTI_CONV8=$m python bench.py --steps 3 --warmup 1 --dtype $dt --no-parity --no-secondary --no-cpu-baseline > $out/conv8_${m}_r${rep}.json 2> $out/conv8_${m}_r${rep}.err
    python - <<PY
import json
d=json.loads(open("$out/conv8_${m}_r${rep}.json").read().strip().splitlines()[-1])
print("CONV8=$m rep $rep %s: %.0f normals/s, conv %.1f ms/step, frac %.4f" % ("$dt", d["value"], d["roofline"]["kernel_ms_per_step"]["conv"], d["roofline"]["frac"]))
PY
  done
done
