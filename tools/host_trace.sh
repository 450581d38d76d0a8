#!/bin/bash
# the host-buffer loop (8-bit frames in, sparse lists out) at depth 2 / 3, fetch on the worker or the main thread;
# copies; a trace of kernels and copies of the default at depth 2:   bash tools/host_trace.sh
export TMPDIR=/tmp
for mode in "2 worker" "2 main" "3 worker" "3 main"; do timeout 300 python3 tools/host_trace.py $mode; done
for mode in "2 worker" "3 worker"; do
  rm -rf gpurun_out/htrace
  timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/htrace -- python3 tools/host_trace.py $mode 2>&1 | grep "ms per step"
  python3 tools/host_trace.py --report gpurun_out/htrace > "gpurun_out/htrace_$(echo $mode | tr ' ' '_').txt"
done
rm -rf gpurun_out/htrace
