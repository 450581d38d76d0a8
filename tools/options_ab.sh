run() { timeout 300 python3 bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$*', '->', round(d['ms_per_step'], 3), 'ms/step', round(d['value'] / 1e6, 1), 'Mkp/s frac', round(r['frac'], 3))"; }
for i in 1 2; do
run
run --set stream_min_waves=256
run --set extrema_stream=1
run --set pyramid_side=0
run --set gate_schedule=0
done
