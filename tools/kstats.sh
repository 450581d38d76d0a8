#!/bin/bash
# per-kernel average durations of the bench under rocprofv3, one line per kernel:  bash tools/kstats.sh [bench args...]
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rm -rf gpurun_out/prof_ks
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ks -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1
f=$(find gpurun_out/prof_ks -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Name"].replace("sift_hip::", "").replace("(anonymous namespace)::", "").replace("void ", ""))
    print("%-44s calls %4s  avg %8.1f us  min %8.1f  max %8.1f  %5s%%" % (name[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
rm -rf gpurun_out/prof_ks
