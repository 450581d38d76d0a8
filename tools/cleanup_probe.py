import os as _os; _os.environ.setdefault("SIFT_HIP_LIBRARY", "libsift_hip_ablate.so")   # measurement options: `make -C sift_amd/csrc ablate`
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sift_amd.sift import Context
ctx = Context(0)
ctx.set_option("diag_cleanup_stamps", 1)
rng = np.random.default_rng(0)
for n, p in [(345000, 0.94), (345000, 0.94)]:
    flags = (rng.random(n) < p).astype(np.uint8)
    out = np.zeros(n, np.int32); cnt = C.c_int32()
    for v in (1,):
        print("n", n, "p", p, "variant", v, flush=True)
        ctx._L.sift_hip_cleanup_survivors(ctx._h, flags, n, out, C.byref(cnt), v)
