#!/bin/bash
# kernel timeline of one bench step under environment settings: tools/env_timeline.sh VAR=VALUE ...
export TMPDIR=/tmp
rm -rf gpurun_out/prof
( for kv in "$@"; do export "$kv"; done
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 > gpurun_out/prof_bench.log 2>&1 )
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 tools/timeline.py "$t"
