#!/bin/bash
export TMPDIR=/tmp
for l in 1 2 4; do
  echo "== lanes $l"
  timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --lanes $l 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step',d['ms_per_step'],'value',d['value'],'roofline',d['roofline']['frac'])"
done
