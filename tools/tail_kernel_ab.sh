#!/bin/bash
# Round 5: the pyramid's tail as one launch (option tail_kernel) against a launch per level on its own stream (tail_kernel=0)
# and in line (tail_async=0), alternately on ONE box.
#   bash tools/tail_kernel_ab.sh [rounds] > gpurun_out/tail_kernel_ab.txt
cd "$(dirname "$0")/.."
rounds=${1:-2}
line() { python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c '
import json, sys
d = json.loads(sys.stdin.read())
r = d["roofline"]
print("%.3f ms/step  repeats %s  frac %.3f  kp %d" % (d["ms_per_step"], " ".join("%.3f" % v for v in d["ms_per_step_repeats"]["all"]), r["frac"], d.get("keypoints_per_step", 0)))'; }
for r in $(seq 1 "$rounds"); do
  for depth in 2 1; do
    echo -n "tail_kernel=1            depth $depth: "; line --pipeline-depth $depth
    echo -n "tail_kernel=0            depth $depth: "; line --pipeline-depth $depth --set tail_kernel=0
    echo -n "tail_async=0             depth $depth: "; line --pipeline-depth $depth --set tail_async=0
  done
done
