#!/bin/bash
# kernels of ONE single-frame calculate() (host frame in, results left on the device)
export TMPDIR=/tmp
cat > /tmp/single_one.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from sift_amd import _lib
from sift_amd.sift import Context, K_SQRT2, pinned_array
from sift_amd.synthetic import synth_frame
img = synth_frame(1920, 1080, 1)
pin = pinned_array((1,) + img.shape, np.float32); pin[0] = img
ctx = Context(0); p = _lib.Params(3, 4, 1.6, K_SQRT2, 0)
for _ in range(6): ctx.calculate_batch(pin, p)
PY
rm -rf gpurun_out/prof1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof1 -- python3 /tmp/single_one.py > /dev/null 2>&1
t=$(find gpurun_out/prof1 -name "*kernel_trace.csv" | head -1); python3 tools/timeline.py "$t" 2>&1 | tail -48
rm -rf gpurun_out/prof1
timeout 120 python3 tools/single_latency.py 2>&1 | grep -v amdgpu
