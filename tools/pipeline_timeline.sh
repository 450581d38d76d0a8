#!/bin/bash
# phase-level kernel timeline of the pipelined bench (two steps in flight), taken from the middle of the run
export TMPDIR=/tmp
rm -rf gpurun_out/prof_p
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_p -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_p.log 2>&1
t=$(find gpurun_out/prof_p -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 tools/timeline_window.py "$t" 8000 20000 > gpurun_out/timeline_pipelined.txt 2>&1
rm -rf gpurun_out/prof_p
cat gpurun_out/timeline_pipelined.txt
