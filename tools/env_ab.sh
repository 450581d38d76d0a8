run() { timeout 300 python3 bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$*', '->', round(d['ms_per_step'], 3), 'ms/step', round(d['value'] / 1e6, 1), 'Mkp/s frac', round(r['frac'], 3))"; }
for i in 1 2; do
run default
HIP_FORCE_DEV_KERNARG=1 run HIP_FORCE_DEV_KERNARG=1
GPU_MAX_HW_QUEUES=6 run GPU_MAX_HW_QUEUES=6
GPU_MAX_HW_QUEUES=12 run GPU_MAX_HW_QUEUES=12
HSA_ENABLE_SDMA=0 run HSA_ENABLE_SDMA=0
done
