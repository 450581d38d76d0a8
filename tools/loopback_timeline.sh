#!/bin/bash
# kernel timeline (all queues) from the middle of `bench.py --rccl-loopback`: where a rank's step goes with the gather on
#   bash tools/loopback_timeline.sh [extra bench args]
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_lb
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_lb -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --rccl-loopback "$@" > gpurun_out/lb_bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/lb_bench.log
t=$(find gpurun_out/prof_lb -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_window.py "$t" ${WINDOW:-11000} ${SKIP:-9000} > gpurun_out/timeline_loopback.txt 2>&1
cat gpurun_out/timeline_loopback.txt
m=$(find gpurun_out/prof_lb -name "*memory_copy_trace.csv" | head -1)
[ -n "$m" ] && python3 - "$m" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.Counter(); n = collections.Counter()
for r in rows:
    k = r.get("Direction", r.get("Kind", "?"))
    acc[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n[k] += 1
for k in acc: print("copies", k, n[k], "total us", acc[k] / 1e3)
PY
rm -rf gpurun_out/prof_lb
