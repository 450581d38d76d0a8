#!/bin/bash
export SIFT_HIP_LIBRARY=libsift_hip_diag.so   # measurement options (desc_dbg, orient_dbg, diag_*, stream_waves): make -C sift_amd/csrc diag
# A/B of the streaming blur's wave count (library option stream_waves; 0 = tile kernel only)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "convolve or pipeline_parity or batch or full_size" -p no:cacheprovider 2>&1 | tail -2
for wv in ${WAVES:-0 1024 2048 3072 4096 8192}; do
  echo "== stream_waves=$wv"
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --set stream_waves=$wv 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step',round(d['ms_per_step'],3),'roofline',round(d['roofline']['achieved']),round(d['roofline']['frac'],3),'blur_ms_total',round(d['roofline']['avg_launch_ms']*16,3))"
done
