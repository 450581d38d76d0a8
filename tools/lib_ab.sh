#!/bin/bash
# A/B of two BUILDS of the library on ONE box under tools/ab.py's protocol (>= 5 alternations, median and min - max per arm, no
# winner when the intervals overlap): sift_amd/lib/libsift_hip.so against the file named by $1 in sift_amd/lib/ (e.g. the previous
# commit's build kept as libsift_hip_base.so).   bash tools/lib_ab.sh libsift_hip_base.so [rounds]
cd "$(dirname "$0")/.."
other=${1:?file name of the other library in sift_amd/lib/}; rounds=${2:-5}
exec python3 tools/ab.py --lib "$(basename "$other")" --rounds "$rounds"
