#!/bin/bash
# Round 5: the pyramid's tail (octaves 2 - 3 of the bench batch) off the main stream (option tail_async) against the launches in
# line, alternately on ONE box; then, with the measurement build, the tail not run at all (upper bound of what hiding it can win).
#   bash tools/tail_ab.sh [rounds] > gpurun_out/tail_ab.txt
cd "$(dirname "$0")/.."
rounds=${1:-3}
bash tools/option_ab.sh tail_async 1 0 "$rounds"
export SIFT_HIP_LIBRARY=libsift_hip_diag.so
line() { python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c '
import json, sys
d = json.loads(sys.stdin.read())
print("%.3f ms/step  repeats %s" % (d["ms_per_step"], " ".join("%.3f" % v for v in d["ms_per_step_repeats"]["all"])))'; }
for depth in 2 1; do
  echo -n "diag build, tail skipped (results wrong)  depth $depth: "; line --pipeline-depth $depth --set diag_skip_tail=1
  echo -n "diag build, tail_async=1                  depth $depth: "; line --pipeline-depth $depth
  echo -n "diag build, tail_async=0                  depth $depth: "; line --pipeline-depth $depth --set tail_async=0
done
