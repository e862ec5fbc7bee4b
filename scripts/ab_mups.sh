#!/bin/bash
# same-box A/B of library variants under .ab/ on BASELINE config 1 (bench.py --mups-only)
for lib in "$@"; do
  cp .ab/$lib nesti-net_amd/libnesti_hip.so
  python bench.py --mups-only --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$lib', round(d['value']), 'frac', round(r['frac'],4), r.get('achieved'), r.get('kernel_ms_per_step'))"
done
