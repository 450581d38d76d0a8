"""Host-buffer loop (8-bit frames in, sparse lists out, BatchPipeline) for a trace of copies and kernels:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/htrace -- python3 tools/host_trace.py [depth] [worker|main]
    python3 tools/host_trace.py --report gpurun_out/htrace"""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def report(d):
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].split("(")[0].replace("void sift_hip::", "").replace("sift_hip::", "")
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + n[:44], r.get("Queue_Id", "")))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"C {r.get('Direction', '')} {r.get('Name', '')}"[:48], ""))
    ev.sort()
    if not ev:
        print("no events")
        return
    # the last third of the run: steady state
    t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * 2 // 3
    t0 = None
    for a, b, n, q in ev:
        if a < t_lo:
            continue
        t0 = a if t0 is None else t0
        if (b - a) < 30000 and n.startswith("K"):
            continue   # small kernels
        print(f"{(a - t0) / 1e3:9.1f} {(b - t0) / 1e3:9.1f} {(b - a) / 1e3:8.1f}  {n} {q}")
        if (a - t0) > 14e6:
            break


if len(sys.argv) > 2 and sys.argv[1] == "--report":
    report(sys.argv[2])
    sys.exit(0)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
from sift_amd import _lib  # noqa: E402
from sift_amd.pipeline import BatchPipeline  # noqa: E402
from sift_amd.sift import K_SQRT2, pinned_array  # noqa: E402
from sift_amd.synthetic import synth_frame  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
where = sys.argv[2] if len(sys.argv) > 2 else "worker"
extra = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in sys.argv[3:])
frames = np.stack([synth_frame(1920, 1080, s + 1) for s in range(8)] * 4).astype(np.uint8)
pin = pinned_array(frames.shape, np.uint8)
pin[...] = frames
params = _lib.Params(3, 4, 1.6, K_SQRT2, 0)
cap = 900000
with BatchPipeline(0, depth=depth, options={"wire_count": 1, **extra}) as pipe:
    bufs = [(pinned_array((cap, 34), np.uint8), pinned_array((cap * 64,), np.float32)) for _ in range(depth)]

    def fetch(c, slot):
        c.results_sparse(bufs[slot][0], bufs[slot][1])

    def loop(n):
        pend = []
        t0 = time.perf_counter()
        for i in range(n + depth):
            if i < n:
                pend.append(pipe.submit(pin, params, then=fetch if where == "worker" else None))
            if pend and (len(pend) >= depth or i >= n):
                tk = pend.pop(0)
                c = tk.result()
                if where != "worker":
                    fetch(c, tk.slot)
                tk.release()
        return (time.perf_counter() - t0) / n * 1e3
    loop(3)
    print(f"depth {depth}, fetch on the {where} thread, {extra or 'default options'}: {loop(12):.2f} ms per step", flush=True)
