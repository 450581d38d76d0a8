#!/bin/bash
# HBM-side traffic and L2 hit rate per kernel (separate passes): FETCH_SIZE (x2 on gfx950 for wide reads), WRITE_SIZE, TCC_HIT/TCC_MISS
export TMPDIR=/tmp
mkdir -p gpurun_out
ARGS="${BENCH_ARGS:---steps 1 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1}"
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  d=gpurun_out/pmck_$(echo $c | cut -d' ' -f1); rm -rf $d
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py $ARGS > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/pmck_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sift_hip::", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[(name, r["Counter_Name"])] += 1
for name in sorted(acc, key=lambda k: -acc[k].get("FETCH_SIZE", 0)):
    a = {k: v / n[(name, k)] for k, v in acc[name].items()}
    rd, wr = a.get("FETCH_SIZE", 0) * 1024 * 2, a.get("WRITE_SIZE", 0) * 1024
    hit, miss = a.get("TCC_HIT_sum", 0), a.get("TCC_MISS_sum", 0)
    print(f"{name[:60]:60s} read(x2) {rd/1e6:9.1f} MB  write {wr/1e6:9.1f} MB  L2 hit {hit/(hit+miss) if hit+miss else 0:.3f}  launches {n[(name,'FETCH_SIZE')]}")
PY
find gpurun_out/pmck_* -name "*.csv" -size +2M -delete
