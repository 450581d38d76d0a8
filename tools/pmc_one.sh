#!/bin/bash
# SQ counters of the kernels whose name matches $1, with bench options $2...:  bash tools/pmc_one.sh descriptor_ --set desc_kernel=2
export TMPDIR=/tmp
pat=$1; shift
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD"; do
  rm -rf gpurun_out/pmc_one
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_one -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 "$@" > /dev/null 2>&1
  python3 - "$pat" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_one/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sift_hip::", "")
        if sys.argv[1] in name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in acc.items():
    n = max(len(v) for v in d.values())
    print(name, "launches", n)
    for k, v in sorted(d.items()):
        print(f"   {k:28s} first launch {v[0]:.4g}   mean {sum(v) / len(v):.4g}")
PY
done
rm -rf gpurun_out/pmc_one
