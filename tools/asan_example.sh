#!/bin/bash
# the C++ multi-GPU example against the ASan build of the host code (build/asan -> sift_amd/lib/asan), until it crashes
set -u
g++ -std=c++17 -pthread -g -fsanitize=address -Iinclude examples/sift_multi_gpu.cpp -Lsift_amd/lib/asan -lsift_hip -Wl,-rpath,$PWD/sift_amd/lib/asan -L/opt/rocm/lib -lamdhip64 -o /tmp/smg_asan || exit 1
for i in $(seq 1 ${N:-40}); do
  ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0 timeout 300 /tmp/smg_asan tests/golden/parrot_r.pgm 5 2 > /tmp/a.out 2> /tmp/a.err; rc=$?
  if [ $rc -ne 0 ]; then echo "run $i exit $rc"; grep -v amdgpu.ids /tmp/a.err | head -60 | cut -c1-300; break; fi
done
echo "done $i"
