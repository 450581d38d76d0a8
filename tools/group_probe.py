"""Timing of sift_hip_group on one GPU: three shards on one device (copies), then one shard with the RCCL loopback."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sift_amd import _lib
from sift_amd.sift import Group, K_SQRT2
from sift_amd.synthetic import synth_frame

params = _lib.Params(3, 3, 1.6, K_SQRT2, 0)
frames = np.stack([synth_frame(320, 240, 40 + i) for i in range(7)])
t0 = time.perf_counter()
g = Group([0, 0, 0])
print(f"create 3 shards: {time.perf_counter() - t0:.3f} s", flush=True)
for wire in (1, 2):
    g.set_option("gather_wire", wire)
    for it in range(4):
        t0 = time.perf_counter()
        g.calculate_batch(frames, params)
        dt = time.perf_counter() - t0
        print(f"wire {wire} batch {it}: {dt * 1e3:.1f} ms, timing {tuple(round(v, 2) for v in g.timing())} exposed {g.gather_exposed_ms():.2f}", flush=True)
t0 = time.perf_counter()
g.submit(frames, params)
for it in range(6):
    g.submit(frames, params)
    g.collect()
g.collect()
print(f"7 pipelined batches: {(time.perf_counter() - t0) * 1e3 / 7:.1f} ms each", flush=True)
t0 = time.perf_counter(); g.close(); print(f"close: {time.perf_counter() - t0:.3f} s", flush=True)

t0 = time.perf_counter()
g = Group([0])
g.set_option("gather_loopback", 1)
g.set_option("gather_transport", 2)
print(f"create: {time.perf_counter() - t0:.3f} s", flush=True)
t0 = time.perf_counter()
print("transport:", g.transport(), f"{time.perf_counter() - t0:.3f} s", flush=True)
for it in range(4):
    t0 = time.perf_counter()
    g.calculate_batch(frames, params)
    print(f"loopback batch {it}: {(time.perf_counter() - t0) * 1e3:.1f} ms, timing {tuple(round(v, 2) for v in g.timing())}", flush=True)
t0 = time.perf_counter(); g.close(); print(f"close: {time.perf_counter() - t0:.3f} s", flush=True)
