#!/bin/bash
# the host-buffer loop (8-bit frames in, sparse lists out) with the transfers as small kernels (default) and as the runtime's
# copies (io_kernels=0); a trace of kernels and copies of the default at depth 2:   bash tools/host_trace.sh
export TMPDIR=/tmp
for mode in "2 worker" "2 main" "3 worker" "3 main" "2 worker io_kernels=0" "2 main io_kernels=0" "3 worker io_kernels=0" "3 main io_kernels=0"; do timeout 300 python3 tools/host_trace.py $mode; done
for mode in "2 worker" "3 worker"; do
  rm -rf gpurun_out/htrace
  timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/htrace -- python3 tools/host_trace.py $mode 2>&1 | grep "ms per step"
  python3 tools/host_trace.py --report gpurun_out/htrace > "gpurun_out/htrace_$(echo $mode | tr ' ' '_').txt"
done
rm -rf gpurun_out/htrace
