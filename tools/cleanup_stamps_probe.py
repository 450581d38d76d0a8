import os, sys
os.environ.setdefault("SIFT_HIP_LIBRARY", "libsift_hip_ablate.so")
sys.path.insert(0, os.getcwd())
import numpy as np
from sift_amd import _lib
from sift_amd.sift import Context, K_SQRT2
from sift_amd.synthetic import synth_frame
img = synth_frame(1920, 1080, 1)
ctx = Context(0); p = _lib.Params(3, 4, 1.6, K_SQRT2, 0)
ctx.set_option("diag_cleanup_stamps", 1)
for _ in range(4):
    ctx.calculate_batch(img[None], p)
print("stages:", {s: ctx.stage(s, 0).size for s in ("candidates", "after_sort1", "after_orient", "after_sort2", "final")})
