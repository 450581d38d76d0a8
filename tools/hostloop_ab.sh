#!/bin/bash
# bench host loop: every context's thread feeds itself (stream) against one dispatching thread (dispatch):  bash tools/hostloop_ab.sh
run() { timeout 300 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$*', '->', round(d['ms_per_step'], 3), 'ms/step', round(d['value'] / 1e6, 1), 'Mkp/s frac', round(r['frac'], 3), 'kp/step', d['config']['keypoints_per_step'])"; }
for s in stream dispatch stream dispatch stream dispatch; do run --host-loop $s; done
