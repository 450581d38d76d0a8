#!/bin/bash
# A/B of the phase gate's schedules (phase_gate.h): bench line for each + the pipelined timeline of schedule 1.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for sch in 0 1 0 1; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --set gate_schedule=$sch > gpurun_out/gate_s$sch.json 2> gpurun_out/gate_s$sch.err
  python3 -c "
import json; d=json.load(open('gpurun_out/gate_s$sch.json')); print('schedule $sch: ms/step', round(d['ms_per_step'],3), 'Mkp/s', round(d['value']/1e6,1), 'frac', round(d['roofline']['frac'],3))" || tail -3 gpurun_out/gate_s$sch.err
done
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --set gate_schedule=1 > /dev/null 2>&1
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_window.py "$t" 8000 500 > gpurun_out/timeline_sched1.txt 2>&1
rm -rf gpurun_out/prof
cat gpurun_out/timeline_sched1.txt
