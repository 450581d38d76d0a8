#!/bin/bash
# Cache-side counters of the kernels whose name matches $1, with bench options $2...:  bash tools/pmc_cache.sh descriptor_wave --set desc_kernel=1
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
pat=$1; shift
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum FETCH_SIZE"; do
  rm -rf gpurun_out/pmc_one
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_one -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 "$@" > /dev/null 2>&1
  python3 - "$pat" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_one/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sift_hip::", "")
        if sys.argv[1] in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/pmc_one/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sift_hip::", "")
        if sys.argv[1] in name:
            dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, d in acc.items():
    print(f"{name[:40]:40s} duration {sum(dur[name]) / max(len(dur[name]), 1):.1f} us")
    for k, v in sorted(d.items()):
        print(f"    {k:32s} mean {sum(v) / len(v):.4g}  (n={len(v)})")
PY
done
rm -rf gpurun_out/pmc_one
