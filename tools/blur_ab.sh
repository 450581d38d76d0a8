#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for th in 64 32; do
  echo "== SIFT_BLUR_TH=$th"
  SIFT_BLUR_TH=$th timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "convolve or pipeline_parity" -p no:cacheprovider 2>&1 | tail -2
  SIFT_BLUR_TH=$th timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step',d['ms_per_step'],'value',d['value'],'roofline',d['roofline']['achieved'],d['roofline']['frac'],'avg_launch_ms',d['roofline']['avg_launch_ms'])"
done
