"""Soak test of the gated BatchPipeline: random batch shapes, parameters that make the reference throw, two and three
slots; every batch is compared with the single context's result.  Not part of the test suite (minutes on the GPU)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sift_amd import _lib
from sift_amd.pipeline import BatchPipeline
from sift_amd.sift import Context, PreconditionViolation
from sift_amd.synthetic import synth_frame

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
N = int(os.environ.get("BATCHES", "120"))
shapes = [(160, 120), (200, 160), (320, 250), (322, 251), (640, 480), (1024, 768)]
jobs = []
for b in range(N):
    w, h = shapes[rng.integers(len(shapes))]
    n = int(rng.integers(1, 5))
    dogs, octaves = [(3, 2), (3, 3), (3, 4), (4, 2), (5, 3)][rng.integers(5)]
    jobs.append((np.stack([synth_frame(w, h, int(rng.integers(1, 1000))) for _ in range(n)]), _lib.Params(dogs, octaves, 1.6, 2 ** 0.5, int(rng.integers(2) if w <= 320 else 0))))
ref = Context(0)
want = []
for imgs, prm in jobs:
    try:
        ref.calculate_batch(imgs, prm)
        want.append((ref.counts().copy(),) + tuple(a.copy() for a in ref.results()))
    except (PreconditionViolation, AssertionError) as e:
        want.append(str(e))
for depth, gated in ((2, True), (3, True), (2, False)):
    t0 = time.time()
    got = []
    with BatchPipeline(0, depth=depth, gated=gated) as pipe:
        tickets = []

        def collect(t):
            try:
                c = t.result()
                got.append((c.counts().copy(),) + tuple(a.copy() for a in c.results()))
            except (PreconditionViolation, AssertionError) as e:
                got.append(str(e))
            t.release()

        for imgs, prm in jobs:
            tickets.append(pipe.submit(imgs, prm))
            if len(tickets) == depth:
                collect(tickets.pop(0))
        for t in tickets:
            collect(t)
    bad = 0
    for g, w_ in zip(got, want):
        if isinstance(w_, str) or isinstance(g, str):
            bad += g != w_
        else:
            bad += not (g[0].tolist() == w_[0].tolist() and g[1].tobytes() == w_[1].tobytes() and g[2].tobytes() == w_[2].tobytes())
    print(f"depth {depth} gated {gated}: {len(got)} batches, {sum(isinstance(w_, str) for w_ in want)} throwing, {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
    assert bad == 0 and len(got) == len(jobs)
print("soak ok")
