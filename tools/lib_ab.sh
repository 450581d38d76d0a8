#!/bin/bash
# A/B of two builds of the library on ONE box (boxes of the pool differ by 3 - 4 % in the bench's headline): the bench's
# headline loop, alternately with sift_amd/lib/libsift_hip.so and with the library named by $1 (e.g. a build of the previous
# commit kept as sift_amd/lib/libsift_hip_base.so); SIFT_HIP_LIBRARY is read by sift_amd/_lib.py.
#   bash tools/lib_ab.sh sift_amd/lib/libsift_hip_base.so [rounds] > gpurun_out/lib_ab.txt
cd "$(dirname "$0")/.."
other=${1:?path of the other library}
rounds=${2:-3}
line() { python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c '
import json, sys
d = json.loads(sys.stdin.read())
print("%.3f ms/step  repeats %s  frac %.3f" % (d["ms_per_step"], " ".join("%.3f" % v for v in d["ms_per_step_repeats"]["all"]), d["roofline"]["frac"]))'; }
for r in $(seq 1 "$rounds"); do
  for depth in 2 1; do
    echo -n "this build   depth $depth: "; line --pipeline-depth $depth
    echo -n "other build  depth $depth: "; SIFT_HIP_LIBRARY="$PWD/$other" line --pipeline-depth $depth
  done
done
