#!/bin/bash
# kernel statistics of the bench with the N > 1 gather path looped through RCCL on this GPU (bench.py --rccl-loopback)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --rccl-loopback > gpurun_out/loopback_profiled.json 2> /dev/null
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"].split("(")[0][:70]
    if any(k in n for k in ("blur", "extrema", "gradient", "descriptor", "cleanup", "orient", "w16", "desc_grid", "out_base", "resample")):
        continue
    print(f'{n:72s} calls {int(r["Calls"]):5d}  total {float(r["TotalDurationNs"])/1e3:10.1f} us  avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
rm -rf gpurun_out/prof
