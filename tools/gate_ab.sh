#!/bin/bash
# phase-gate schedules on one box:  bash tools/gate_ab.sh
run() { timeout 600 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$*', '->', round(d['ms_per_step'], 3), 'ms/step', round(d['value'] / 1e6, 1), 'Mkp/s frac', round(r['frac'], 3))"; }
for s in ${SCHEDULES:-1 0 1 0}; do run --set gate_schedule=$s; done
