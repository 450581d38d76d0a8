#!/bin/bash
# DBGS="" (tools/profile_round.sh): only the SQ counters of the shipped library; otherwise the ablation build (options desc_dbg,
# orient_dbg: `make -C sift_amd/csrc ablate`)
[ -n "${DBGS-0 1 2 3}" ] && export SIFT_HIP_LIBRARY=libsift_hip_ablate.so
# descriptor_wave_kernel: time with phases switched off (desc_dbg bits: 1 no neighbour chains, 2 no histograms; WRONG results) and SQ counters
export TMPDIR=/tmp
mkdir -p gpurun_out
for dbg in ${DBGS-0 1 2 3}; do
  rm -rf gpurun_out/prof_p
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 --set desc_dbg=$dbg > /dev/null 2>&1
  f=$(find gpurun_out/prof_p -name "*kernel_stats.csv" | head -1)
  echo "dbg $dbg: $(grep -E 'descriptor_wave|desc_grid' $f | sed 's/(.*)"/"/' | cut -d, -f1-4 | tr '\n' ' ')"
done
rm -rf gpurun_out/prof_p gpurun_out/pmc_d1 gpurun_out/pmc_d2
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_d1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 > /dev/null 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_d2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_d1", "gpurun_out/pmc_d2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sift_hip::", "")
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[(name, r["Counter_Name"])] += 1
    for name in sorted(acc):
        if name.startswith(("descriptor", "desc_grid", "extrema_edge", "gradient4", "orientation_kernel")):
            print(name, {k: f"{v / n[(name, k)]:.4g}" for k, v in sorted(acc[name].items())}, "launches", max(n[(name, k)] for k in acc[name]))
PY
find gpurun_out/pmc_d1 gpurun_out/pmc_d2 -name "*.csv" -size +2M -delete
