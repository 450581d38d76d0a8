#!/bin/bash
# How much of a descriptor-kernel saving reaches the PIPELINED step: the bench's headline loop with parts of the kernel switched off (measurement
# build, timing only, wrong results): desc_dbg 1 = no neighbour chains (the kernel alone: 0.85 -> 0.50 ms), 2 = no histograms (-> 0.76), 15 = nothing.
#   bash tools/sensitivity_probe.sh > gpurun_out/sens.txt
cd "$(dirname "$0")/.."
export SIFT_HIP_LIBRARY="$PWD/sift_amd/lib/libsift_hip_ablate.so"
line() { python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c '
import json, sys
d = json.loads(sys.stdin.read())
print("%.3f ms/step  repeats %s  frac %.3f" % (d["ms_per_step"], " ".join("%.3f" % v for v in d["ms_per_step_repeats"]["all"]), d["roofline"]["frac"]))'; }
for v in 0 1 2 3 15; do echo -n "desc_dbg=$v depth 2: "; line --set desc_dbg=$v; done
echo -n "desc_dbg=0 depth 2: "; line
