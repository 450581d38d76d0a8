#!/bin/bash
# phase-level timeline of the pipelined bench (two batches in flight), without the rest of the profile round
export TMPDIR=/tmp
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_window.py "$t" 8000 12500 > gpurun_out/timeline_pipelined.txt 2>&1
rm -rf gpurun_out/prof
head -70 gpurun_out/timeline_pipelined.txt
