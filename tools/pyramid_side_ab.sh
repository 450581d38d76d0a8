#!/bin/bash
# option pyramid_side (an octave's top level on a stream of its own, beside the next octave's first launches): bench A/B + timeline
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 0 1 0 1; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --set pyramid_side=$v 2> gpurun_out/ps.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[0]); print('pyramid_side $v: ms/step',round(d['ms_per_step'],3),'Mkp/s',round(d['value']/1e6,1),'blur frac',round(d['roofline']['frac'],3))" || tail -3 gpurun_out/ps.err
done
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --pipeline-depth 1 --set pyramid_side=1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[0]); print('pyramid_side 1 depth 1: ms/step',round(d['ms_per_step'],3))"
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 --set pyramid_side=1 > /dev/null 2>&1
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); python3 tools/timeline.py "$t" > gpurun_out/timeline_ps.txt 2>&1
rm -rf gpurun_out/prof
grep -E "blur_|resample|w16|extrema_edge|step span" gpurun_out/timeline_ps.txt | head -30
