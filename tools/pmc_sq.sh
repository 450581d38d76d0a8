#!/bin/bash
# SQ-level counters for the non-pyramid kernels (one pass, kernel-trace only).
set -u
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_sq
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_sq.log 2>&1
echo "exit $?"
rm -rf gpurun_out/pmc_sq2
timeout 900 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_sq2.log 2>&1
echo "exit $?"
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/pmc_sq", "gpurun_out/pmc_sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sift_hip::", "")
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    for name in ("descriptor_kernel", "cleanup1_kernel", "orientation_kernel", "edge_filter_kernel", "gradient_kernel", "blur_stream_kernel<5, true, 4>", "blur_stream_kernel<10, true, 2>", "blur_stream_kernel<7, true, 2>", "extrema_mask_kernel"):
        if name in acc:
            print(name, {k: f"{v:.3g}" for k, v in sorted(acc[name].items())})
    for name in sorted(acc):
        if name.startswith("blur_fused") or name.startswith("extrema_edge") or name.startswith("gradient4"):
            print(name, {k: f"{v:.3g}" for k, v in sorted(acc[name].items())})
PY
find gpurun_out/pmc_sq gpurun_out/pmc_sq2 -name "*.csv" -size +4M -delete
