import os as _os; _os.environ.setdefault("SIFT_HIP_LIBRARY", "libsift_hip_ablate.so")   # measurement options: `make -C sift_amd/csrc ablate`
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sift_amd import _lib
from sift_amd.sift import Context, K_SQRT2
from sift_amd.synthetic import synth_frame
frames = np.stack([synth_frame(1920,1080,s+1) for s in range(8)]*4)
d = torch.from_numpy(frames).cuda()
ctx = Context(0); p = _lib.Params(3,4,1.6,K_SQRT2,0)
for dbg in [0,1,2,3,8,12,4]:
    ctx.set_option("desc_dbg", dbg)
    for _ in range(2): ctx.calculate_batch_device(d.data_ptr(), 32, 1920, 1080, p)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): ctx.calculate_batch_device(d.data_ptr(), 32, 1920, 1080, p)
    torch.cuda.synchronize(); print('dbg',dbg,'ms/step',(time.perf_counter()-t)/5*1e3, flush=True)
