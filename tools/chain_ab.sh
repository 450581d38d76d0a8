#!/bin/bash
# the small octaves as one launch (chain_from=o) against a launch per level (0):  bash tools/chain_ab.sh
run() { timeout 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$*', '->', round(d['ms_per_step'], 3), 'ms/step', round(d['value'] / 1e6, 1), 'Mkp/s frac', round(r['frac'], 3), 'alone', round(r.get('frac_alone') or 0, 3), 'spot', d.get('parity_spot_check'))"; }
for s in ${CHAINS:-2 0 2 0 3 1}; do run --set chain_from=$s; done
