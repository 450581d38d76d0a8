#!/bin/bash
timeout 900 python3 tools/group_probe.py 2>&1 | grep -v amdgpu.ids
g++ -std=c++17 -pthread -Iinclude examples/sift_multi_gpu.cpp -Lsift_amd/lib -lsift_hip -Wl,-rpath,$PWD/sift_amd/lib -L/opt/rocm/lib -lamdhip64 -o /tmp/smg || exit 1
for i in 1 2 3 4 5 6; do timeout 120 /tmp/smg tests/golden/parrot_r.pgm 5 2 2>&1 | tail -2 | cut -c1-250; echo "exit $?"; done
