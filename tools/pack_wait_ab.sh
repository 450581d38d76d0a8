run() { timeout 600 python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads([ln for ln in sys.stdin.read().splitlines() if ln.startswith('{')][-1]); c = d['config']
print('$LABEL $*', '->', round(d['ms_per_step'], 3), 'ms/step', [round(x,3) for x in d['ms_per_step_repeats']['all']], 'gather ms/step', c['gather_ms_per_step'])"; }
for rep in 1 2 3; do
  LABEL="new " run --rccl-loopback
  export SIFT_HIP_LIBRARY=$PWD/sift_amd/lib/libsift_hip_base.so; LABEL="base" run --rccl-loopback; unset SIFT_HIP_LIBRARY
done
LABEL="new N=1" run
for i in 1 2; do python3 tools/gather_probe.py 40 2>&1 | grep "ms/step"; done
