"""Latency of one Sift::calculate() call on a single 1920x1080 frame handed over as a host buffer (the drop-in's
call pattern), results copied back to host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sift_amd.sift import Sift
from sift_amd.synthetic import synth_frame
img = synth_frame(1920, 1080, 1)
s = Sift(3, 4)
for _ in range(3): pts = s.calculate(img)
t = time.perf_counter()
for _ in range(10): pts = s.calculate(img)
dt = (time.perf_counter() - t) / 10
print(f"single 1920x1080 frame, host buffer in, InterestPoint list out: {dt*1e3:.2f} ms, {len(pts)} keypoints")
from sift_amd import _lib
from sift_amd.sift import Context, K_SQRT2
ctx = Context(0); p = _lib.Params(3, 4, 1.6, K_SQRT2, 0)
for _ in range(3): ctx.calculate_batch(img[None], p)
t = time.perf_counter()
for _ in range(10): ctx.calculate_batch(img[None], p); n = ctx.total()
dt = (time.perf_counter() - t) / 10
print(f"C ABI only (host frame in, results left on device): {dt*1e3:.2f} ms, {n} keypoints")
