#!/bin/bash
# A/B on ONE box: for each library variant under .ab/ (plus env overrides), run bench.py and print value/frac.
# usage: scripts/ab_bench.sh "old.so" "new.so" "new.so:SOME_ENV=0" ...
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=${spec#*:}
  cp .ab/$lib nesti-net_amd/libnesti_hip.so
  env $envs python bench.py --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$spec', round(d['value']), round(r['frac'],4), {k:round(v,1) for k,v in r['kernel_ms_per_step'].items()})"
done
