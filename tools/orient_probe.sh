#!/bin/bash
export SIFT_HIP_LIBRARY=libsift_hip_ablate.so   # measurement options (desc_dbg, orient_dbg, diag_*, stream_waves): make -C sift_amd/csrc ablate
# orientation_kernel: time with phases switched off (orient_dbg bits: 1 no peak search, 2 no ordered sums, 4 no window loads; WRONG results)
export TMPDIR=/tmp
mkdir -p gpurun_out
for dbg in ${DBGS-0 1 2 4 3 7}; do
  rm -rf gpurun_out/prof_p
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 --set orient_dbg=$dbg > /dev/null 2>&1
  f=$(find gpurun_out/prof_p -name "*kernel_stats.csv" | head -1)
  echo "dbg $dbg: $(grep -E 'orientation_kernel|orient_emit|orient_count' $f | sed 's/(.*)"/"/' | cut -d, -f1-4 | tr '\n' ' ')"
done
rm -rf gpurun_out/prof_p
