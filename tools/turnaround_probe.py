import os as _os; _os.environ.setdefault("SIFT_HIP_LIBRARY", "libsift_hip_diag.so")   # measurement options: `make -C sift_amd/csrc diag`
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from sift_amd import _lib
from sift_amd.pipeline import BatchPipeline
from sift_amd.synthetic import synth_frame
frames = torch.from_numpy(np.stack([synth_frame(1920, 1080, s) for s in range(1, 33)])).cuda()
p = _lib.Params(3, 4, 1.6, float(np.float32(np.sqrt(2.0))), 0)
for rep in (1, 5, 1, 5, 10):
    with BatchPipeline(0, 2, {"diag_repeat": rep}) as pipe:
        calls = 40 // rep
        def run(k):
            left = [k]
            def source():
                if left[0] <= 0: return None
                left[0] -= 1
                return (frames.data_ptr(), 32, 1920, 1080, p)
            pipe.run_stream(source, None)
        run(2 if rep > 1 else 6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(calls)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"diag_repeat {rep}: {dt / (calls * rep) * 1e3:.3f} ms per batch", flush=True)
