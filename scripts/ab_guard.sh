#!/bin/bash
# Same-box A/B of the conditioning guard (bench.py headline + single-stream pass, 3 steps): guard on (calibrated threshold) vs off,
# and optional library variants under .ab/ (usage: ab_guard.sh [variant.so ...])
cd $GRAFT_REPO_ROOT
run() { # label, lib ("" = in-tree), extra args
  NESTI_LIB=$2 python bench.py --no-cpu-baseline --no-parity --no-secondary --steps 3 --warmup 1 $3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', 'two-stream', round(d['value']), 'single', round(d['single_stream']['value']), 'guard', {k:d.get('x8_guard',{}).get(k) for k in ('rechecked','thr_eff')})"
}
for rnd in 1 2; do
run guard_on "" ""
run guard_off "" "--x8-guard-thr -1"
for v in "$@"; do run $v $GRAFT_REPO_ROOT/.ab/$v ""; done
done
