#!/bin/bash
export SIFT_HIP_LIBRARY=libsift_hip_diag.so   # measurement options (desc_dbg, orient_dbg, diag_*, stream_waves): make -C sift_amd/csrc diag
# streaming blur: waves a launch is cut into (option stream_waves), bench line per setting
export TMPDIR=/tmp
for sw in ${SWS:-2048 3072 4096 2048 4096}; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --set stream_waves=$sw 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[0]); print('stream_waves $sw ms/step',round(d['ms_per_step'],3),'Mkp/s',round(d['value']/1e6,1),'blur frac',round(d['roofline']['frac'],3), 'avg launch ms', round(d['roofline']['avg_launch_ms'],4))"
done
