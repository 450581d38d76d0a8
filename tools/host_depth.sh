#!/bin/bash
# host-buffer boundary (u8 in, sparse out) at pipeline depth 2 and 3:  bash tools/host_depth.sh
mkdir -p gpurun_out/ab
for d in 2 3; do
  timeout 600 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --pmc-traffic 0 --pipeline-depth $d > gpurun_out/ab/hd$d.json 2> gpurun_out/ab/hd$d.err
  python3 - $d <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/ab/hd{sys.argv[1]}.json").read().strip().splitlines()[-1])
h = d["host_inclusive"]
print("depth", sys.argv[1], round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in h.items() if not k.endswith("what")}, "single frame", round(d.get("single_frame_ms", 0), 3))
PY
done
