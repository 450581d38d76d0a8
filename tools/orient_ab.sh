#!/bin/bash
export SIFT_HIP_LIBRARY=libsift_hip_ablate.so   # measurement options (desc_dbg, orient_dbg, diag_*, stream_waves): make -C sift_amd/csrc ablate
export TMPDIR=/tmp
for d in 0 1 2 4 7; do
  rm -rf gpurun_out/prof_o
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_o -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --set orient_dbg=$d > /dev/null 2>&1
  f=$(find gpurun_out/prof_o -name "*kernel_stats.csv" | head -1)
  echo "dbg $d: $(grep orientation_kernel $f | sed "s/.*)\",//" | cut -d, -f1-3) cleanup1: $(grep cleanup1_kernel $f | sed 's/.*)",//' | cut -d, -f1-3)"
done
rm -rf gpurun_out/prof_o
