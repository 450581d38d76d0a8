#!/bin/bash
export SIFT_HIP_LIBRARY=libsift_hip_diag.so   # measurement options (desc_dbg, orient_dbg, diag_*, stream_waves): make -C sift_amd/csrc diag
# single-step kernel timeline with the gradient pass serialised behind the extrema pass (every kernel alone on the chip)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 --set diag_serial_gradient=1 ${1:-} > /dev/null 2>&1
t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); python3 tools/timeline.py "$t" > gpurun_out/timeline_serial.txt 2>&1
rm -rf gpurun_out/prof
grep -vE "blur_|resample" gpurun_out/timeline_serial.txt
