#!/bin/bash
# HBM traffic of the kernels from the PMC counters (separate passes, MI355X_MICROARCH.md §HBM):
# FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/pmc_$c.log 2>&1
  echo "pmc $c exit $?"
done
python3 tools/pmc_summary.py gpurun_out > gpurun_out/pmc_summary.json
cat gpurun_out/pmc_summary.json | head -60
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -size +8M -delete
