"""PCIe-inclusive rate: the boundary handed HOST buffers (sift_hip_calculate_batch)
and the results copied back to host memory, versus device-resident input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sift_amd import _lib
from sift_amd.sift import Context, K_SQRT2
from sift_amd.synthetic import synth_frame
frames = np.stack([synth_frame(1920, 1080, s + 1) for s in range(8)] * 4)
ctx = Context(0); p = _lib.Params(3, 4, 1.6, K_SQRT2, 0)
pinned = torch.from_numpy(frames).pin_memory().numpy()
d = torch.from_numpy(frames).cuda()
def run(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): n = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3, n
def dev():
    ctx.calculate_batch_device(d.data_ptr(), 32, 1920, 1080, p); return ctx.total()
def host_pageable():
    ctx.calculate_batch(frames, p); kp, desc = ctx.results(); return kp.size
def host_pinned():
    ctx.calculate_batch(pinned, p); kp, desc = ctx.results(); return kp.size
for name, fn in (("device-resident input, results left on device", dev), ("pageable host input + results to host", host_pageable),
                 ("pinned host input + results to host", host_pinned)):
    ms, n = run(fn)
    print(f"{name}: {ms:.2f} ms per 32-frame batch, {n / ms * 1e3 / 1e6:.1f} M keypoints/s", flush=True)

# the same through the BatchPipeline: slot k+1's worker uploads while slot k computes and its results are read back
from sift_amd.pipeline import BatchPipeline
for depth in (2, 3):
    with BatchPipeline(0, depth=depth) as pipe:
        def stream(reps, src):
            tickets, n = [], 0
            for _ in range(reps):
                tickets.append(pipe.submit(src, p))
                if len(tickets) == depth:
                    t = tickets.pop(0); c = t.result(); kp, desc = c.results(); n = kp.size; t.release()
            for t in tickets:
                c = t.result(); kp, desc = c.results(); n = kp.size; t.release()
            return n
        for label, src in (("pageable", frames), ("pinned", pinned)):
            stream(3, src)
            t0 = time.perf_counter(); n = stream(10, src); ms = (time.perf_counter() - t0) / 10 * 1e3
            print(f"BatchPipeline depth {depth}, {label} host input + results to host: {ms:.2f} ms per 32-frame batch, {n / ms * 1e3 / 1e6:.1f} M keypoints/s", flush=True)
