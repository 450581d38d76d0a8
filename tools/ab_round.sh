#!/bin/bash
# A/B of library options on one box: bench line per option set (ms/step, roofline), then the per-launch table of the default.
#   bash tools/ab_round.sh "" "pyramid_side=0" "stream_min_waves=512"
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
i=0
for opt in "$@"; do
  args=""
  for kv in $opt; do args="$args --set $kv"; done
  timeout 600 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extras $args > gpurun_out/ab/b$i.json 2> gpurun_out/ab/b$i.err
  python3 - "$opt" gpurun_out/ab/b$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"[{sys.argv[1] or 'default'}] {d['ms_per_step']:.3f} ms/step  {d['value'] / 1e6:.1f} Mkp/s  frac {r['frac']:.3f}  (sum-based {r['frac_over_sum_of_durations']:.3f})  busy {r['avg_launch_ms'] * 16 * 1e3:.0f} us/pyramid")
except Exception as e:
    print(f"[{sys.argv[1]}] failed: {e}")
PY
  i=$((i+1))
done
if [ "${TRACE:-1}" = "1" ]; then
  rm -rf gpurun_out/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras ${TRACE_ARGS:-} > gpurun_out/ab/prof_bench.log 2>&1
  t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1)
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$t" ] && python3 tools/roofline_from_stats.py --trace "$t" | tee gpurun_out/ab/roofline.txt
  [ -n "$f" ] && cp "$f" gpurun_out/ab/kernel_stats.csv && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'].split('(')[0][-60:]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Percentage']:>6s}%")
PY
  [ -n "$t" ] && python3 - "$t" <<'PY'
import csv, sys, collections
# per kernel and position within a step: average duration (a kernel launched k times per step shows k lines)
rows = [(r["Kernel_Name"].split("(")[0].replace("void sift_hip::", "").replace("sift_hip::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: r[1])
by = collections.defaultdict(list)
for n, a, b in rows:
    by[n].append(b - a)
steps = 10
for n, d in by.items():
    if len(d) % steps or "blur" in n:
        continue
    k = len(d) // steps
    if k > 1 and k <= 8:
        print(f"{n[:50]:50s} per step: " + "  ".join(f"{sum(d[i::k]) / steps / 1e3:7.1f}" for i in range(k)) + " us")
PY
  find gpurun_out/prof -name "*kernel_trace.csv" -delete
fi
