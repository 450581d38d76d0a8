#!/bin/bash
# The C++ multi-GPU example N times on the HIP runtime of /opt/rocm (7.2) and N times on the one PyTorch bundles (7.0.2, the runtime every
# Python host of the library - bench.py, the tests - runs on once torch is imported): exit statuses of both.
#   N=150 bash tools/example_loop_runtime.sh > gpurun_out/example_loop_runtime.txt
cd "$(dirname "$0")/.."
g++ -std=c++17 -pthread -Iinclude examples/sift_multi_gpu.cpp -Lsift_amd/lib -lsift_hip -Wl,-rpath,$PWD/sift_amd/lib -L/opt/rocm/lib -lamdhip64 -o /tmp/smg || exit 1
T=$(python3 -c "import os, importlib.util as u; print(os.path.join(os.path.dirname(u.find_spec('torch').origin), 'lib'))")
for rt in system torch; do
  ok=0; declare -A st=()
  for i in $(seq 1 ${N:-150}); do
    if [ $rt = torch ]; then LD_PRELOAD=$T/libamdhip64.so timeout 120 /tmp/smg tests/golden/parrot_r.pgm 5 2 > /tmp/smg.out 2> /tmp/smg.err; rc=$?
    else timeout 120 /tmp/smg tests/golden/parrot_r.pgm 5 2 > /tmp/smg.out 2> /tmp/smg.err; rc=$?; fi
    if [ $rc -eq 0 ] && grep -q "^ok: 5 frames over 2 shards" /tmp/smg.out; then ok=$((ok+1)); else st[$rc]=$(( ${st[$rc]:-0} + 1 )); fi
  done
  echo -n "$rt runtime: ok $ok of ${N:-150}"; for k in "${!st[@]}"; do echo -n "; exit $k x ${st[$k]}"; done; echo
  unset st
done
