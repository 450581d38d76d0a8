#!/usr/bin/env python3
"""Headline benchmark: keypoints/sec of the Sift::calculate() hot path on synthetic 1920x1080
greyscale frames, 4 octaves x 3 DoGs (BASELINE.json metric / config 4's per-GPU share).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (launched by torch.distributed.run for N > 1).  A "step" is one pass of the
whole hot path (pyramid, DoG, extrema, edge filter, orientation, descriptors) over one batch of
FRAMES_PER_GPU device-resident frames per GPU; for N > 1 every step's keypoint lists (records +
descriptors, never images) are gathered on rank 0 over RCCL, the transfer of step k overlapping the
kernels of step k+1, all inside the timed region.  Two steps are in flight per GPU (--pipeline-depth 2: two
contexts joined by a phase gate, sift_amd/csrc/phase_gate.h): step k+1's extrema / gradient pass runs under step
k's cleanup, which cannot fill the chip; pyramids never share the chip, so the roofline figure measured on the blur
launches is that of the kernel alone.  Every step is complete inside the timed region.  Weak scaling: per-GPU work
is fixed.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from multiprocessing.pool import ThreadPool

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Two batches in flight per GPU use 2 contexts x 2 streams (+ torch's): with the HIP runtime's default of 4 hardware
# queues, streams of different contexts share a queue and pick up each other's ordering.  Must be set before the
# runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FRAMES_PER_GPU = 32          # BASELINE config 4: 256 frames over 8 GPUs
W, H = 1920, 1080
DOGS, OCTAVES, SIGMA = 3, 4, 1.6
SUBPIXEL = 0
# name -> (w, h, dogs, octaves, subpixel, frames per GPU, label)
WORKLOADS = {
    "config4": (1920, 1080, 3, 4, 0, 32, "batch of {n} synthetic 1920x1080 greyscale frames per GPU, sigma 1.6, k sqrt2, "
                "4 octaves x 3 DoGs, subpixel off (BASELINE config 4 per-GPU share)"),
    "config3": (1920, 1080, 3, 4, 1, 8, "batch of {n} synthetic 1920x1080 frames per GPU, subpixel=1 (3840x2160 base), 4 octaves x 3 DoGs: "
                "BASELINE config 3 at its nearest non-throwing parameters (4 oct x 5 DoG throws in the reference, App. B-13)"),
    "config5": (3840, 2160, 3, 5, 1, 8, "batch of {n} synthetic 3840x2160 frames per GPU, subpixel=1 (7680x4320 base), 5 octaves x 3 DoGs: "
                "BASELINE config 5 at its nearest non-throwing parameters (6 octaves throws in the reference, App. B-14)"),
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
CPU_SAMPLE_FRAMES = 8


def cpu_baseline(frames):
    """Oracle (lean mode: identical results to the reference, redundant copies hoisted) on a
    bounded sample of the same workload, 1 thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    kps, secs = 0, 0.0
    for f in frames:
        run = O.OracleRun(f, DOGS, OCTAVES, SIGMA)
        secs += run.seconds
        kps += run.points("final")[0].size
        run.close()
    ref = None
    try:   # what the reference's own binary needed for frame 1 of this workload (measured once in the build container)
        pin = np.load(os.path.join(ROOT, "tests", "golden", "refpin_bench_frame.npz"))
        ref = {"value": float(pin["points"].size / float(pin["seconds"])), "unit": "keypoints/s", "cores": 1,
               "seconds_per_frame": float(pin["seconds"]), "keypoints": int(pin["points"].size),
               "how": "Sift::calculate of the reference's prebuilt binary, called in-process through oracle/refexec in the build "
                      "container (not on this box: the reference does not travel); same keypoints and descriptors, bit for bit"}
    except Exception:
        pass
    return {"reference_binary": ref, "value": kps / secs, "unit": "keypoints/s", "cores": 1, "kind": "port",
            "sample": f"{len(frames)} of the {FRAMES_PER_GPU} synthetic 1920x1080 frames, 4 oct x 3 DoG, oracle in lean mode "
                      f"(reference's per-candidate image copies and per-keypoint re-blur hoisted; same results), "
                      f"{secs:.1f} s CPU, {kps} keypoints"}


def pmc_traffic():
    """HBM bytes per blur launch from the committed rocprofv3 PMC pass (tools/pmc_round.sh; FETCH_SIZE x2
    corrected + WRITE_SIZE, MI355X_MICROARCH.md section HBM).  bench.py itself cannot read PMC counters."""
    p = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
    try:
        return json.load(open(p))["_blur_fused_all"]["hbm_bytes_per_launch"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--workload", default="config4", choices=sorted(WORKLOADS),
                    help="config4 (default) is the headline workload; config3 / config5 are BASELINE.json's other GPU "
                         "configurations at their nearest non-throwing parameters (SURVEY 8(d)), reported as labelled extra lines")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "exercise the N > 1 flow where ranks have to share one GPU: results are staged through host memory)")
    ap.add_argument("--share-device", action="store_true", help="testing: every rank uses GPU 0")
    ap.add_argument("--wire", default="sparse", choices=["sparse", "packed", "full"],
                    help="N > 1, descriptors on the wire (all lossless): sparse = presence bits + the floats that are set (about a third), "
                         "packed = the 112 floats that can carry information, full = 128 floats")
    ap.add_argument("--set", action="append", default=[], metavar="OPTION=VALUE", help="library option (sift_hip_set_option), e.g. fused_edge=0")
    ap.add_argument("--pipeline-depth", type=int, default=2,
                    help="batches in flight per GPU (sift_amd.pipeline.BatchPipeline: one context and host thread per slot).\n"
                         "2 (default): consecutive steps overlap under the phase gate - the next step's extrema / gradient pass\n"
                         "fills the chip while this step's cleanup (one workgroup per image) cannot, and no pyramid shares\n"
                         "the chip, so the per-launch roofline figure is that of the kernel alone.  1: one step at a time")
    ap.add_argument("--pipeline-gate", type=int, default=1, choices=[0, 1],
                    help="pipeline depth > 1: 1 (default) joins the contexts with a phase gate (no pyramid shares the chip); 0 leaves the interleaving to the GPU's queues")
    args = ap.parse_args()

    global W, H, DOGS, OCTAVES, SUBPIXEL
    W, H, DOGS, OCTAVES, SUBPIXEL, frames_default, label = WORKLOADS[args.workload]
    if args.frames is None:
        args.frames = frames_default
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")

    import torch
    import torch.distributed as dist
    from sift_amd import _lib
    from sift_amd.sift import Context, K_SQRT2
    from sift_amd.synthetic import synth_frame

    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm_dev = dev if args.backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    nf = args.frames
    seeds = [rank * nf + i + 1 for i in range(nf)]  # config 4: seeds 1..256 block-sharded
    with ThreadPool(min(8, os.cpu_count() or 1)) as pool:
        frames = np.stack(pool.map(lambda s: synth_frame(W, H, s), seeds))
    d_frames = torch.from_numpy(frames).to(dev)
    torch.cuda.synchronize()

    from sift_amd.pipeline import BatchPipeline

    depth = max(1, args.pipeline_depth)
    pipe = BatchPipeline(local_rank, depth, dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in args.set), gated=bool(args.pipeline_gate))
    ctxs = pipe.contexts
    ctx = ctxs[0]
    params = _lib.Params(DOGS, OCTAVES, SIGMA, K_SQRT2, SUBPIXEL)
    L = ctx._L

    from sift_amd.gather import device_results, gather_finish, gather_start

    # N > 1: the RCCL gather of step k (keypoint records + descriptors to rank 0, never images) is only
    # STARTED at the end of step k and overlaps the kernels of step k+1, which run on the library's own
    # streams; two result buffers alternate, and every gather is finished inside the timed region.
    in_flight = []          # (GatherHandle, tensors kept alive)
    tickets = []            # submitted steps whose results have not been collected yet (at most depth - 1 between steps)

    def collect(ticket):
        """Finish one step: wait for its batch, hand its keypoint lists to the gather (N > 1), free its slot."""
        c = ticket.result()
        total = c.total()
        if world > 1:
            while len(in_flight) > 1:      # at most two gathers in flight
                gather_finish(in_flight.pop(0)[0])
            # The library's result arrays are read in place (no staging copy); packing / cloning them is queued on
            # torch's stream right away and is long done when the slot's next descriptor kernel rewrites them.
            # wire format (lossless, sift_amd/gather.py): bytes per record, floats per descriptor (None: as many as are set)
            bpk, fpk = {"sparse": (34, None), "packed": (20, 112), "full": (20, 128)}[args.wire]
            kp, desc = device_results(c, total, dev, wire=args.wire)
            counts = torch.from_numpy(c.counts()).to(comm_dev)
            if comm_dev.type == "cpu":     # test backend: stage through host memory
                kp, desc = kp.cpu(), desc.cpu()
            in_flight.append((gather_start(kp, desc, counts, dst=0, floats_per_kp=fpk, bytes_per_kp=bpk), (kp, desc)))
        ticket.release()
        return total

    def step():
        """Submit one pass over the batch; collect the oldest step once `depth` are in flight."""
        tickets.append(pipe.submit_device(d_frames.data_ptr(), nf, W, H, params))
        return collect(tickets.pop(0)) if len(tickets) >= depth else 0

    def drain():
        total = 0
        while tickets:
            total += collect(tickets.pop(0))
        while in_flight:
            gather_finish(in_flight.pop(0)[0])
        return total

    for _ in range(args.warmup):
        step()
    drain()
    for c in ctxs:
        c.set_option("profile", 1)
        c.profile_reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    kps = 0
    for _ in range(args.steps):
        kps += step()
    kps += drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.set_option("profile", 0)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        k = torch.tensor([kps], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
        kps = int(k.item())

    if rank == 0:
        prof = [c.profile(0) for c in ctxs]
        ms, launches, nbytes = (sum(p[i] for p in prof) for i in range(3))
        achieved = (nbytes / 1e9) / (ms / 1e3) if ms > 0 else 0.0
        out = {
            "metric": "keypoints/sec, 1920x1080 4oct/3DoG" if args.workload == "config4" else f"keypoints/sec, {args.workload} (not the headline metric)",
            "value": kps / dt,
            "unit": "keypoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": label.format(n=nf),
                       "frames_per_gpu": nf, "frames_total": nf * world, "pipeline_depth": depth, "pipeline_gate": bool(args.pipeline_gate) and depth > 1, "keypoints_per_step": kps // max(args.steps, 1),
                       "frames_per_s": nf * world * args.steps / dt,
                       "gather": ("RCCL p2p of keypoint records + descriptors to rank 0, started per step and overlapped with the next step; "
                                  + {"full": "128 floats per descriptor", "packed": "descriptors on the wire as 112 of 128 floats (bin 7 of each cell is structurally +0.0f; lossless)",
                                     "sparse": "descriptors on the wire as 112 presence bits + the floats that are not +0.0f (about a third; lossless)"}[args.wire]) if world > 1 else "none (1 GPU)"},
            "roofline": {"kernel": "blur_stream_kernel / blur_fused_kernel (separable Gaussian + DoG; every launch of the pyramid)",
                         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic() if (args.workload == "config4" and nf == FRAMES_PER_GPU and (depth == 1 or args.pipeline_gate)) else None,
                         "traffic_source": "profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench)",
                         "timing": "hipEventElapsedTime over start/stop events attached to each blur dispatch (hipExtLaunchKernelGGL) on the library's stream, inside the timed region",
                         "launches": launches, "avg_launch_ms": ms / launches if launches else None,
                         "algorithmic_bytes_per_launch": nbytes / launches if launches else None},
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "config4":
            out["cpu_baseline"] = cpu_baseline(frames[:CPU_SAMPLE_FRAMES])
        print(json.dumps(out), flush=True)
    pipe.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
