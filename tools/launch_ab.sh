#!/bin/bash
# Round 5: does taking the runtime's per-launch lookup of the host stub out of the launch path (sift_amd/csrc/launch_cache.h:
# function objects resolved once per device, hipExtModuleLaunchKernel) end the crashes of multi-threaded C++ hosts?
# examples/sift_multi_gpu.cpp (three host threads launching on one GPU + the gather thread) against two builds of the library,
# alternately in blocks on ONE box:  A = this build (module launch),  B = sift_amd/lib_static (-DSIFT_HIP_STATIC_LAUNCH: the
# runtime's own hipLaunchKernel path, everything else identical).
#   make -C sift_amd/csrc OUT=../lib_static OBJ=../../build/obj_static DIAGFLAG=-DSIFT_HIP_STATIC_LAUNCH ../lib_static/libsift_hip.so
#   bash tools/launch_ab.sh [blocks=3] [runs per block=50] > gpurun_out/launch_ab.txt
cd "$(dirname "$0")/.."
blocks=${1:-3}; per=${2:-50}
for v in lib lib_static; do
  g++ -std=c++17 -pthread -Iinclude examples/sift_multi_gpu.cpp -Lsift_amd/$v -lsift_hip -Wl,-rpath,$PWD/sift_amd/$v -L/opt/rocm/lib -lamdhip64 -o /tmp/smg_$v || exit 1
done
gcc -shared -fPIC -o /tmp/segv_bt.so tools/probe/segv_bt.c
declare -A ok bad
for v in lib lib_static; do ok[$v]=0; bad[$v]=0; done
for b in $(seq 1 "$blocks"); do
  for v in lib lib_static; do
    for i in $(seq 1 "$per"); do
      LD_PRELOAD=/tmp/segv_bt.so timeout 120 /tmp/smg_$v tests/golden/parrot_r.pgm 5 2 > /tmp/smg.out 2> /tmp/smg.err; rc=$?
      if [ $rc -eq 0 ]; then ok[$v]=$((ok[$v]+1)); else bad[$v]=$((bad[$v]+1)); echo "block $b $v run $i: exit $rc"; grep -v amdgpu.ids /tmp/smg.err | head -12 | cut -c1-200; fi
    done
    echo "after block $b: module launch (lib) ok ${ok[lib]} bad ${bad[lib]} | static launch (lib_static) ok ${ok[lib_static]} bad ${bad[lib_static]}"
  done
done
