"""Where the N > 1 per-rank step loses time against N = 1, on one GPU: the bench's stream host loop (two gated contexts, each
thread feeding itself) with the per-batch work of the gather switched on piece by piece:
   A  nothing (the N = 1 loop)                       B  + wire_count (the descriptor kernel counts the sparse format's floats)
   C  + the pack, queued on the side stream, lists dropped
   D  + a gather thread that waits for each pack (no transport)
   E  + the transport: KeypointGather(loopback=True) through RCCL to this same rank (what bench.py --rccl-loopback runs)
   python tools/gather_probe.py [steps]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sift_amd import _lib                                     # noqa: E402
from sift_amd.gather import GatherThread, KeypointGather, device_results   # noqa: E402
from sift_amd.pipeline import BatchPipeline                   # noqa: E402
from sift_amd.sift import K_SQRT2                             # noqa: E402
from sift_amd.synthetic import synth_frame                    # noqa: E402


class _NoTransport:
    def __init__(self):
        self._rec_lock = threading.Lock()

    def records_buffer(self, n):
        return torch.empty(n, dtype=torch.uint8, device="cuda:0")

    def push(self, *a):
        return []

    def flush(self):
        return []


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    nf, W, H = 32, 1920, 1080
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    frames = np.stack([synth_frame(W, H, s + 1) for s in range(nf)])
    d_frames = torch.from_numpy(frames).to(dev)
    params = _lib.Params(3, 4, 1.6, K_SQRT2, 0)
    item = (d_frames.data_ptr(), nf, W, H, params)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29547")
    import torch.distributed as dist
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    probe = torch.ones(1, device=dev)
    dist.all_reduce(probe)

    def run(label, wire_count, pack, thread, transport):
        pipe = BatchPipeline(0, 2, {"wire_count": 1} if wire_count else {}, gated=True)
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for rep in range(2):     # first repetition warms up
            n = steps if rep else 6
            gatherer = (KeypointGather(nf, dev, dst=0, loopback=True) if transport else _NoTransport()) if pack else None
            gt = GatherThread(gatherer, None, cuda_device=0) if thread else None
            left, seq_next, seq_of = [n], [0], {}

            def source():
                if left[0] <= 0:
                    return None
                left[0] -= 1
                seq_of[threading.get_ident()] = seq_next[0]
                seq_next[0] += 1
                return item

            def sink(c, slot, _i):
                if not pack:
                    return
                total = c.total()
                with torch.cuda.stream(streams[slot]):
                    rec_out = gatherer.records_buffer(total * 34)
                    kp, desc = device_results(c, total, dev, wire="sparse", rec_out=rec_out, defer_pack=True)
                if gt is not None:
                    gt.put(seq_of[threading.get_ident()], kp, desc, c.counts().copy(), ready=c.pack_wait)

            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.run_stream(source, sink)
            if gt is not None:
                gt.close()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{label:70s} {dt / steps * 1e3:7.3f} ms/step", flush=True)
        pipe.close()

    for _ in range(2):
        run("A  N = 1 loop", False, False, False, False)
        run("B  + wire_count", True, False, False, False)
        run("C  + pack queued on the side stream, lists dropped", True, True, False, False)
        run("D  + gather thread waiting for the packs, no transport", True, True, True, False)
        run("E  + RCCL loopback transport", True, True, True, True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
