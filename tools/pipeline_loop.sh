#!/bin/bash
# examples/sift_pipeline.cpp (two Sift objects, two threads, one gate) many times
g++ -std=c++17 -pthread -Iinclude examples/sift_pipeline.cpp -Lsift_amd/lib -lsift_hip -Wl,-rpath,$PWD/sift_amd/lib -L/opt/rocm/lib -lamdhip64 -o /tmp/spl || exit 1
gcc -shared -fPIC -o /tmp/segv_bt.so tools/probe/segv_bt.c
ok=0; bad=0
for i in $(seq 1 ${N:-100}); do
  LD_PRELOAD=/tmp/segv_bt.so timeout 120 /tmp/spl tests/golden/parrot_r.pgm ${FRAMES:-6} > /tmp/spl.out 2> /tmp/spl.err; rc=$?
  if [ $rc -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "run $i: exit $rc"; grep -v amdgpu.ids /tmp/spl.err | head -${LINES_SHOWN:-30} | cut -c1-200; fi
done
echo "pipeline example: ok $ok bad $bad"
