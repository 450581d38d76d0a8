#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "convolve or pipeline_parity or batch or full_size" -p no:cacheprovider 2>&1 | tail -2
for cfg in "4 2048" "2 2048" "2 4096" "2 6144" "0 2048" "0 4096"; do
  set -- $cfg
  echo "== SIFT_STREAM_CPL=$1 SIFT_STREAM_WAVES=$2"
  SIFT_STREAM_CPL=$1 SIFT_STREAM_WAVES=$2 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step',round(d['ms_per_step'],3),'roofline',round(d['roofline']['achieved']),round(d['roofline']['frac'],3),'blur_ms_total',round(d['roofline']['avg_launch_ms']*16,3))"
done
