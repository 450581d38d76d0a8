#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprof kernel stats.  Logs land in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt
lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/device.txt
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py --steps ${STEPS:-3} --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"; tail -c 2500 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
timeout 600 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 > gpurun_out/bench_depth1.log 2>/dev/null; echo "bench depth 1 exit $?"; grep -o "ms_per_step\": [0-9.]*" gpurun_out/bench_depth1.log
if [ "${PROF:-1}" = "1" ]; then
  rm -rf gpurun_out/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_bench.log 2>&1
  echo "rocprof exit $?"
  find gpurun_out/prof -name "*kernel_stats*" | head -3
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200
  # keep the merge-back small: drop the per-dispatch trace
  find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
  t=$(find gpurun_out/prof -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 tools/timeline_window.py "$t" 8000 500 > gpurun_out/timeline_pipelined.txt 2>&1
  # single-step timeline (one step at a time)
  rm -rf gpurun_out/prof1
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --pipeline-depth 1 > gpurun_out/prof1_bench.log 2>&1
  t=$(find gpurun_out/prof1 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 tools/timeline.py "$t" > gpurun_out/timeline.txt 2>&1
  rm -rf gpurun_out/prof1
fi
