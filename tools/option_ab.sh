#!/bin/bash
# A/B of one library option on ONE box: the bench's headline loop with OPTION=A and OPTION=B alternately, two batches in flight and one.
#   bash tools/option_ab.sh lazy_top 1 0 [rounds] > gpurun_out/option_ab.txt
cd "$(dirname "$0")/.."
opt=${1:?option}; a=${2:?value A}; b=${3:?value B}; rounds=${4:-3}
line() { python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c '
import json, sys
d = json.loads(sys.stdin.read())
r = d["roofline"]
print("%.3f ms/step  repeats %s  frac %.3f  avg launch %.4f ms" % (d["ms_per_step"], " ".join("%.3f" % v for v in d["ms_per_step_repeats"]["all"]), r["frac"], r.get("avg_launch_ms", 0)))'; }
for r in $(seq 1 "$rounds"); do
  for depth in 2 1; do
    for v in $a $b; do
      echo -n "$opt=$v  depth $depth: "; line --pipeline-depth $depth --set $opt=$v
    done
  done
done
