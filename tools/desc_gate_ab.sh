#!/bin/bash
# descriptor kernel x gate variants on one box:  bash tools/desc_gate_ab.sh
run() { timeout 600 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$*', '->', round(d['ms_per_step'], 3), 'ms/step', round(d['value'] / 1e6, 1), 'Mkp/s frac', round(r['frac'], 3))"; }
for rep in 1 2; do
for dk in ${KERNELS:-2 1}; do
  for opt in "" "--set gate_mid=1" "--set gate_mid=2" "--set gate_schedule=0"; do run --set desc_kernel=$dk $opt; done
done
done
