#!/bin/bash
# Does the descriptor kernel's window traffic (3.4 GB of L2 misses per bench batch) cost the PIPELINED step anything?  The measurement build's
# desc_dbg bits: 4 = no window loads at all, 128 = every window from the image's first rows (same instructions, cache hits), 8 = 16-byte aligned
# window loads (wrong pixels, fewer lines touched).  Results are wrong with any of them; timing only.
#   bash tools/desc_traffic_probe.sh > gpurun_out/desc_traffic_probe.txt
cd "$(dirname "$0")/.."
export SIFT_HIP_LIBRARY="$PWD/sift_amd/lib/libsift_hip_ablate.so"
line() { python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python3 -c '
import json, sys
d = json.loads(sys.stdin.read())
print("%.3f ms/step  repeats %s  frac %.3f" % (d["ms_per_step"], " ".join("%.3f" % v for v in d["ms_per_step_repeats"]["all"]), d["roofline"]["frac"]))'; }
for r in 1 2; do
  for depth in 2 1; do
    for v in 0 128 4 8; do
      echo -n "desc_dbg=$v  depth $depth: "; line --pipeline-depth $depth --set desc_dbg=$v
    done
  done
done
