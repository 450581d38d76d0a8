#!/bin/bash
# One box: the orientation / gradient parity tests, two bench lines (two batches in flight, one at a time) and the
# single-step kernel timeline.   bash tools/quick_check.sh ["-k expr"]
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -m gpu -x -q ${1:-} 2>&1 | tail -4
for depth in 2 1 2; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --pipeline-depth $depth 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth $depth ms/step',round(d['ms_per_step'],3),'Mkp/s',round(d['value']/1e6,1),'blur frac',round(d['roofline']['frac'],3))"
done
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 > /dev/null 2>&1
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); python3 tools/timeline.py "$t" > gpurun_out/timeline_single.txt 2>&1
rm -rf gpurun_out/prof
cat gpurun_out/timeline_single.txt
