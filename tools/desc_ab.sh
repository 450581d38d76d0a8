#!/bin/bash
# A/B of the descriptor kernels on one box: tile per wave (desc_kernel=2) vs wave per keypoint (1), bench line and kernel stats.
#   KERNELS="2 1" STEPS=10 bash tools/desc_ab.sh
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for dk in ${KERNELS:-2 1}; do
  for depth in 1 2; do
    echo "== desc_kernel=$dk depth=$depth"
    timeout 600 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-extras --pipeline-depth $depth --set desc_kernel=$dk 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step',round(d['ms_per_step'],3),'Mkp/s',round(d['value']/1e6,1),'blur frac',round(d['roofline']['frac'],3))"
  done
done
done
for dk in ${KERNELS:-2 1}; do
  rm -rf gpurun_out/prof_d$dk
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_d$dk -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --pipeline-depth 1 --set desc_kernel=$dk > /dev/null 2>&1
  f=$(find gpurun_out/prof_d$dk -name "*kernel_stats.csv" | head -1)
  echo "-- desc_kernel=$dk"; grep -E "descriptor|desc_cell|cleanup2|out_base" $f | sed 's/(.*)"/"/' | cut -c1-120
  find gpurun_out/prof_d$dk -name "*kernel_trace.csv" -delete
done
