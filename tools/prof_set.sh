#!/bin/bash
# rocprofv3 kernel timeline of the bench with library options: tools/prof_set.sh OPTION=VALUE ...
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof
args=(); for kv in "$@"; do args+=(--set "$kv"); done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 "${args[@]}" > gpurun_out/prof_bench.log 2>&1
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 tools/timeline.py "$t" > gpurun_out/timeline.txt 2>&1
cat gpurun_out/timeline.txt
