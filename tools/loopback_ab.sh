#!/bin/bash
# The N > 1 per-rank step on one GPU (keypoint lists looped through RCCL to the same rank) against the plain N = 1 step, both
# host loops, alternating on one box:  bash tools/loopback_ab.sh
run() { timeout 600 python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads([ln for ln in sys.stdin.read().splitlines() if ln.startswith('{')][-1]); r = d['roofline']; c = d['config']
print('$*', '->', round(d['ms_per_step'], 3), 'ms/step', round(d['value'] / 1e6, 1), 'Mkp/s frac', round(r['frac'], 3), 'gather ms/step', c['gather_ms_per_step'], 'arrived', c['gather_steps_on_rank0'])"; }
for rep in 1 2 3; do
  run
  run --rccl-loopback
  run --rccl-loopback --host-loop dispatch
done
