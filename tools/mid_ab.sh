#!/bin/bash
# descriptors of batch g released at octave o of the next pyramid (gate_mid=o) instead of after it (0):  bash tools/mid_ab.sh
run() { timeout 600 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$*', '->', round(d['ms_per_step'], 3), 'ms/step', round(d['value'] / 1e6, 1), 'Mkp/s frac', round(r['frac'], 3))"; }
for s in ${MIDS:-0 2 3 0 2 3}; do run --set gate_mid=$s; done
