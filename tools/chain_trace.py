import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from sift_amd import _lib
from sift_amd.sift import Context
from sift_amd.synthetic import synth_frame
frames = np.stack([synth_frame(1920, 1080, s) for s in range(1, 33)])
ctx = Context(0)
ctx.set_option("chain_from", int(os.environ.get("CHAIN_FROM", "2")))
ctx.set_option("chain_mode", int(os.environ.get("CHAIN_MODE", "1")))
for kv in filter(None, os.environ.get("SIFT_SET", "").split(",")):   # SIFT_SET="name=value,name=value": any library option
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
if "CHAIN_SPREAD" in os.environ:
    ctx.set_option("chain_spread", int(os.environ["CHAIN_SPREAD"]))
p = _lib.Params(3, 4, 1.6, float(np.float32(np.sqrt(2.0))), 0)
for _ in range(4):
    ctx.calculate_batch(frames, p)
ctx.close()
