#!/bin/bash
# rocprofv3 kernel stats of the bench (no tests); env passes through
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 > gpurun_out/prof_bench.log 2>&1
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -${LINES_OUT:-30} "$f" | cut -c1-${CUT:-150}
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 tools/timeline.py "$t" > gpurun_out/timeline.txt 2>&1
