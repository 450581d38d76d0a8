#!/bin/bash
# the C++ multi-GPU example many times: any crash with its whole stderr
g++ -std=c++17 -pthread -Iinclude examples/sift_multi_gpu.cpp -Lsift_amd/lib -lsift_hip -Wl,-rpath,$PWD/sift_amd/lib -L/opt/rocm/lib -lamdhip64 -o /tmp/smg || exit 1
gcc -shared -fPIC -o /tmp/segv_bt.so tools/probe/segv_bt.c
ok=0; bad=0
for i in $(seq 1 ${N:-30}); do
  LD_PRELOAD=/tmp/segv_bt.so timeout 120 /tmp/smg tests/golden/parrot_r.pgm ${FRAMES:-5} ${SHARDS:-2} > /tmp/smg.out 2> /tmp/smg.err; rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "run $i: exit $rc"; tail -5 /tmp/smg.out | cut -c1-300; grep -v amdgpu.ids /tmp/smg.err | head -${LINES_SHOWN:-60} | cut -c1-300; fi
done
echo "ok $ok bad $bad"
