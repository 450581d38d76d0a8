"""The level chain (chain_from=2; chain_mode 1 = agent-scope accesses, 2 = the same with every image's tiles spread over all XCDs,
0 = ordinary accesses + fences) against a launch per level: two DIFFERENT batches alternate on one context, so that a level read
stale (from the previous batch's buffers) cannot go unnoticed.   timeout 120 python3 tools/chain_probe.py [n] [w] [h] [reps]"""
import sys, time, hashlib
import numpy as np
sys.path.insert(0, ".")
from sift_amd import _lib
from sift_amd.sift import Context

n, w, h, reps = (int(a) for a in (sys.argv[1:5] + ["8", "512", "384", "6"][len(sys.argv) - 1:]))
rng = np.random.default_rng(3)
def batch():
    imgs = (rng.random((n, h, w), dtype=np.float32) * 255).astype(np.float32)
    for _ in range(2):   # smooth a little so that there are keypoints
        imgs = (imgs + np.roll(imgs, 1, 1) + np.roll(imgs, 1, 2) + np.roll(imgs, -1, 1) + np.roll(imgs, -1, 2)) / 5
    return imgs
batches = [batch(), batch()]
params = _lib.Params(3, 4, 1.6, float(np.float32(np.sqrt(2.0))), 0)
ref = None
for mode in (None, 1, 2, 0):
    ctx = Context(0)
    ctx.set_option("chain_from", 0 if mode is None else 2)
    if mode is not None:
        ctx.set_option("chain_mode", mode)
    got, ms = [], []
    try:
        for rep in range(reps):
            t0 = time.perf_counter()
            ctx.calculate_batch(batches[rep % 2], params)
            ms.append((time.perf_counter() - t0) * 1e3)
            kp, de = ctx.results()
            got.append((int(ctx.counts().sum()), hashlib.sha256(kp.tobytes() + de.tobytes()).hexdigest()[:16]))
        if ref is None:
            ref = got
        print("chain", "off" if mode is None else f"mode {mode}", got[:2], "ms", round(min(ms), 3),
              "identical to per-level launches:", all(g == ref[i % 2] for i, g in enumerate(got)), flush=True)
    except Exception as e:
        print("chain mode", mode, "FAILED:", e, flush=True)
    ctx.close()
