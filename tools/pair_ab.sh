#!/bin/bash
# The first two pyramid levels as one launch (option blur_pair) against two, and how many waves that launch is cut into, on ONE box.
#   bash tools/pair_ab.sh [rounds] > gpurun_out/pair_ab.txt
cd "$(dirname "$0")/.."
rounds=${1:-2}
line() { python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c '
import json, sys
d = json.loads(sys.stdin.read())
print("%.3f ms/step  repeats %s" % (d["ms_per_step"], " ".join("%.3f" % v for v in d["ms_per_step_repeats"]["all"])))'; }
for r in $(seq 1 "$rounds"); do
  for depth in 2 1; do
    echo -n "two launches          depth $depth: "; line --pipeline-depth $depth --set blur_pair=0
    for pw in 1536 2048 2560 3072 4096; do
      echo -n "one launch, $pw waves depth $depth: "; line --pipeline-depth $depth --set blur_pair=1 --set pair_waves=$pw
    done
  done
done
