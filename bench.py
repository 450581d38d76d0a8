#!/usr/bin/env python3
"""Headline benchmark: keypoints/sec of the Sift::calculate() hot path on synthetic 1920x1080
greyscale frames, 4 octaves x 3 DoGs (BASELINE.json metric / config 4's per-GPU share).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (launched by torch.distributed.run for N > 1).  A "step" is one pass of the
whole hot path (pyramid, DoG, extrema, edge filter, orientation, descriptors) over one batch of
FRAMES_PER_GPU device-resident frames per GPU; for N > 1 every step's keypoint lists (records +
descriptors, never images) are gathered on rank 0 over RCCL, the transfer of step k overlapping the
kernels of step k+1, all inside the timed region.  Two steps are in flight per GPU (--pipeline-depth 2: two
contexts joined by a phase gate, sift_amd/csrc/phase_gate.h): step k's cleanup chain, which cannot fill the chip, runs
under step k+1's pyramid and its descriptors under step k+1's extrema / gradient pass.  `roofline.frac` is measured on
the blur launches of the timed region, i.e. beside that chain; `roofline.frac_alone` is the same measurement on a few
steps run one at a time after the timed region (the blur launches alone on the chip).  At N = 1 each of the two contexts'
host threads takes its next step itself (--host-loop stream, BatchPipeline.run_stream; `dispatch` is the single submitting
thread of rounds 1 - 3, which N > 1 still uses because the gather is pushed from that thread).  Every step is complete inside the
timed region.  Weak scaling: per-GPU work is fixed.  Rank 0 prints ONE JSON line.
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
GATHER_SETTLE_STEPS = 16
PROFILE_EVERY = 4            # the events cost ~10 us per blur launch (0.16 ms per step): sample
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
CPU_SAMPLE_FRAMES = 8


def host_cpu():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"host_cores": os.cpu_count(), "host_cpu": model}


def _result_digest(pts, desc):
    """SHA-256 over one image's returned keypoints (x, y, octave, index, scale, orientation) and its 128-float descriptors."""
    import hashlib
    h = hashlib.sha256()
    for f in ("x", "y", "octave", "index", "scale", "orientation"):
        h.update(np.ascontiguousarray(pts[f]).tobytes())
    h.update(np.ascontiguousarray(desc, dtype=np.float32).tobytes())
    return h.hexdigest()


def cpu_baseline(frames, faithful=True):
    """Oracle on a bounded sample of the same workload, 1 thread, on this box's host cores: lean mode (identical results to
    the reference, its redundant copies hoisted) on 8 frames, and faithful mode (the reference's own cost structure: three
    DoG images copied per extremum candidate, sift.cpp:297-298, a level re-blurred per keypoint, sift.cpp:87) on one."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    kps, secs, per_frame = 0, 0.0, []
    for f in frames:
        run = O.OracleRun(f, DOGS, OCTAVES, SIGMA)
        secs += run.seconds
        pts, desc = run.points("final")
        kps += pts.size
        per_frame.append((int(pts.size), _result_digest(pts, desc)))
        run.close()
    faith = None
    if faithful:
        # bounded: the cost is O(candidates x pixels), a whole 1080p frame takes this mode 456 s on the MI355X box's EPYC 9575F
        # (profiles/r02_bench_full.json; 19764 keypoints: 43 keypoints/s) - its top-left 960x540 quarter ~25 s there (round 6; until
        # round 5 the sample was the 480x270 sixteenth: 1.5 s, 750 - 800 keypoints/s, 18 x the whole frame's rate)
        h, w = frames[0].shape
        crop = np.ascontiguousarray(frames[0][:h // 2, :w // 2])
        run = O.OracleRun(crop, DOGS, OCTAVES, SIGMA, faithful=True)
        n = run.points("final")[0].size
        faith = {"value": n / run.seconds, "unit": "keypoints/s", "cores": 1, "seconds": run.seconds, "keypoints": int(n),
                 "sample": f"the top-left {w // 2}x{h // 2} quarter of frame 1, oracle in faithful mode (the reference's own cost structure: "
                           "three DoG images copied per extremum candidate, a level re-blurred per keypoint), timed on this box; the cost "
                           "grows with candidates x pixels, so a whole 1080p frame is still ~4 x slower per keypoint (measured once on "
                           "this box type: 456 s for frame 1, 43 keypoints/s, profiles/r02_bench_full.json)"}
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
    lean = {"value": kps / secs, "unit": "keypoints/s", "cores": 1, "kind": "port",
            "sample": f"{len(frames)} of the {FRAMES_PER_GPU} synthetic 1920x1080 frames, 4 oct x 3 DoG, oracle in lean mode "
                      f"(reference's per-candidate image copies and per-keypoint re-blur hoisted; same results), "
                      f"{secs:.1f} s CPU, {kps} keypoints"}
    out = {"per_frame": per_frame, "reference_binary": ref, "lean_port": lean, **host_cpu()}
    if faith is not None:
        # the headline baseline is the REFERENCE'S cost structure (what `Sift::calculate` costs on this host), not the faster port
        out.update({"value": faith["value"], "unit": "keypoints/s", "cores": 1, "kind": "port", "sample": faith["sample"],
                    "seconds": faith["seconds"], "keypoints": faith["keypoints"],
                    "whole_frame_measured_once": {"value": 43.3, "unit": "keypoints/s", "seconds": 456.0, "keypoints": 19764,
                                                  "what": "the same mode on the whole of frame 1 (1920x1080), this box type, round 2: profiles/r02_bench_full.json"}})
    else:
        out.update(lean)
    return out


def pmc_traffic_live(extra_args):  # noqa: C901
    """HBM bytes per blur launch, measured now: two child runs of this script (one step each) under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes; FETCH_SIZE x2 for 16-byte-per-lane reads on gfx950,
    MI355X_MICROARCH.md section HBM), averaged over the blur launches.  None when rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    tot, launches = {}, 0
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, c)
            cmd = [exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--pipeline-depth", "1"] + extra_args
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            except Exception as ex:   # noqa: BLE001
                return None, f"rocprofv3 pass {c}: {ex}"
            if r.returncode != 0:
                return None, f"rocprofv3 pass {c} exited {r.returncode}"
            s, n = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    # every launch of the family the roofline is about: streaming / tile blurs, the kept-pixels reductions, the pair launch
                    if row.get("Counter_Name") == c and any(k in row["Kernel_Name"] for k in ("blur_stream_kernel", "blur_stream2_kernel", "blur_fused_kernel", "blur_reduce_kernel", "blur_pair_kernel")):
                        s += float(row["Counter_Value"])
                        n += 1
            if not n:
                return None, f"no blur launches in the {c} pass"
            tot[c], launches = s / n, n
    return tot["FETCH_SIZE"] * 1024 * 2 + tot["WRITE_SIZE"] * 1024, f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this bench on this box ({launches} blur launches each; FETCH_SIZE x2)"


def launch_ranks(n):
    """Start `n` children of this script, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
    torch.distributed.run would), wait for all, return the exit status (non-zero if any rank failed).  Called before anything in
    this process has initialised the GPU; the children are ordinary subprocesses (no exec of a process that touched HIP).
    Rank 0 inherits stdout, so its one JSON line is this command's one JSON line; the other ranks' stdout goes to stderr."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=_JSON_FD if r == 0 else sys.stderr))   # (this process's fd 1 points at stderr: see the end of the file)
    status = 0
    try:
        pending = dict(enumerate(procs))
        while pending:
            for r, p in list(pending.items()):
                rc = p.poll()
                if rc is None:
                    continue
                del pending[r]
                if rc != 0:
                    print(f"bench.py: rank {r} exited with status {rc}", file=sys.stderr)
                    status = status or (rc if rc > 0 else 1)
                    for q in pending.values():     # a dead rank leaves the others waiting in a collective: end them (exact PIDs)
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return status


# FP32 vector peak of the data sheet, 157.3 TFLOP/s, counts a packed FMA as 4: 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = 78.6 T
# lane-operations/s where a multiply and an add are two operations (the reference's unfused arithmetic, -ffp-contract=off).
# What the issue logic sustains was measured (tools/probe/valu_rate_probe.hip, profiles/r05_valu_rate.txt): v_pk_mul_f32 /
# v_pk_add_f32 streams reach 59 T at two waves per SIMD (where the streaming blurs run: 140 - 240 VGPRs), 64 at three, 72 at
# eight.  (Rounds 4 - 5 priced this roof at 39.3 T - 64 lanes per clock and CU - which was a factor of two too low.)
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
VALU_ISSUE_CEILING_2_WAVES = 59.0


def blur_lane_ops(nf, W, H, octaves, dogs, sigma, k):
    """Lane-operations (FP32 multiplies + adds, the reference's unfused arithmetic) of one batch's pyramid as the kernels do it:
    a level blur of radius R costs 2 (2R+1) per pixel in the row pass and (R+1) + (2R+1) in the column pass (one product serves
    two slots: kernels_pyramid.hip) = 7R + 4; a kept-pixels reduction (2R+1) + (3R+2) / 4 per SOURCE pixel (kernels_reduce.hip).
    Radii: (int)(3 sigma + 0.5) of Sift::_createDOGs' scale schedule (sift.cpp:388-411)."""
    def radius(s):
        return max(1, int(3.0 * float(np.float32(s)) + 0.5))
    ops, exp = 0.0, 0
    w, h = W, H
    prev_scale = sigma
    ops += w * h * (7 * radius(sigma) + 4)                      # g(0,0)
    for o in range(octaves):
        scales = []
        for j in range(1, dogs + 1):
            scales.append(float(np.float32((k ** exp) * sigma)))
            exp += 1
        for sc in scales:
            ops += w * h * (7 * radius(sc) + 4)
        if o < octaves - 1:
            r = radius(scales[dogs - 2])                        # reduceToNextLevel(g(o, D-1), its scale)
            ops += w * h * ((2 * r + 1) + (3 * r + 2) / 4.0)
            exp -= 2
            w, h = (w + 1) // 2, (h + 1) // 2
    return ops * nf


def whole_step_fraction(nf, W, H, keypoints, seconds):
    """All stages' algorithmic bytes of one step (4 octaves x 3 DoGs) over the step's time, against the 8 TB/s peak, with SURVEY.md
    section 8(d)'s per-unit figures: pyramid 4 B read + 4 B written per pixel and level (round 5: Gaussian levels only, three per
    octave; a reduction reads its source and writes a quarter), gradient maps 12 B per pixel of level (0,0), extremum scan 16 B
    per pixel of every octave (it reads four Gaussian levels and forms the DoGs itself; 12 B when it read three DoG levels and the
    pyramid wrote them: 7.87 GB per step in all, 7.54 now), descriptor stages 3.5 KB per keypoint."""
    px = [W * H] + [((W + (1 << o) - 1) >> o) * ((H + (1 << o) - 1) >> o) for o in range(1, 4)]
    pyramid = 0.0
    for o in range(4):
        pyramid += px[o] * (8 if o == 0 else 0)          # g(0,0) from the input
        pyramid += px[o] * (8 + 8 + 8)                   # g(o,1), g(o,2), g(o,3): no DoG level is written (the extremum scan forms them)
        if o < 3:
            pyramid += px[o] * 4 + px[o + 1] * 4         # reduceToNextLevel: source read, kept pixels written
    gradient = px[0] * 12
    extrema = sum(px) * 16
    descriptors = keypoints / nf * 3584
    total = nf * (pyramid + gradient + extrema + descriptors)
    return {"algorithmic_gbytes_per_step": total / 1e9, "achieved": total / 1e9 / seconds, "frac": total / 1e9 / seconds / HBM_PEAK_GBS,
            "what": "algorithmic bytes of ALL stages of a step (pyramid, gradient maps, extremum scan, 3.5 KB per keypoint for the descriptor stages: "
                    "SURVEY.md section 8(d)) over ms_per_step, against the same 8 TB/s"}


def dropin_cpp_leg(frame, iterations=50):
    """The reference's ACTUAL boundary, timed outside the timed region: examples/sift_dropin_bench.cpp calls
    sift::Sift::calculate(Image2f&) (include/sift/sift.hpp: float host image in, std::vector<InterestPoint> with a heap
    std::vector<f32_t> per point out - /root/reference/sift.hpp:78, interestpoint.hpp:46, main.cpp:52-57) `iterations` times
    on frame 1 of the batch from one thread, then from two gated Sift objects on two threads.  A child process of its own
    (started, not exec'ed); a host without g++ or a failing run reports the reason instead of figures."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "sift_amd", "lib", "sift_dropin_bench")
    src = os.path.join(ROOT, "examples", "sift_dropin_bench.cpp")
    hdr = os.path.join(ROOT, "include", "sift", "sift.hpp")
    try:
        if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), src,
                                   "-L" + os.path.join(ROOT, "sift_amd", "lib"), "-lsift_hip",
                                   "-Wl,-rpath," + os.path.join(ROOT, "sift_amd", "lib"), "-o", exe], timeout=300)
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "frame.f32")
            np.ascontiguousarray(frame, np.float32).tofile(path)
            h, w = frame.shape
            r = subprocess.run([exe, path, str(w), str(h), str(iterations)], capture_output=True, text=True, timeout=600,
                               env=dict(os.environ, GPU_MAX_HW_QUEUES="8"))
        if r.returncode != 0:
            return {"error": f"exit status {r.returncode}: {r.stderr[-300:]}"}
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)["dropin_cpp"]
        d["what"] = ("sift::Sift::calculate(Image2f&) of include/sift/sift.hpp in a C++ host (examples/sift_dropin_bench.cpp): pageable float "
                     "frame in, std::vector<InterestPoint> with one heap vector of 128 floats per keypoint out, the result dropped again "
                     "(ms_per_frame = call + drop, mean); two_gated_objects: two Sift objects joined by a gate, one host thread each")
        return d
    except Exception as e:   # noqa: BLE001 - a leg beside the headline must not take the line with it
        return {"error": repr(e)[:300]}


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
    ap.add_argument("--profile-every", type=int, default=PROFILE_EVERY, help="every N-th batch of a context carries the per-launch timing events of the roofline figure")
    ap.add_argument("--host-loop", choices=("stream", "dispatch"), default="stream",
                    help="'stream' - each context's host thread takes its next step itself (N > 1: and packs its lists for the gather thread); "
                         "'dispatch' - one thread submits, collects and pushes to the gather (rounds 1 - 3)")
    ap.add_argument("--no-extras", action="store_true", help="skip the legs beside the headline: host-buffer rate, single-frame latency, live PMC traffic")
    ap.add_argument("--pmc-traffic", type=int, default=1, choices=[0, 1],
                    help="1 (default): roofline.traffic is measured live by two child passes of this bench under rocprofv3 --pmc "
                         "(FETCH_SIZE / WRITE_SIZE, one step each); 0: traffic stays null")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "exercise the N > 1 flow where ranks have to share one GPU: results are staged through host memory)")
    ap.add_argument("--share-device", action="store_true", help="testing: every rank uses GPU 0")
    ap.add_argument("--wire", default="sparse", choices=["sparse", "packed", "full"],
                    help="N > 1, descriptors on the wire (all lossless): sparse = presence bits + the floats that are set (about a third), "
                         "packed = the 112 floats that can carry information, full = 128 floats")
    ap.add_argument("--set", action="append", default=[], metavar="OPTION=VALUE", help="library option (sift_hip_set_option), e.g. pair_waves=2048; SIFT_HIP_LIBRARY=libsift_hip_diag.so adds the forcing options")
    ap.add_argument("--pipeline-depth", type=int, default=2,
                    help="batches in flight per GPU (sift_amd.pipeline.BatchPipeline: one context and host thread per slot).\n"
                         "2 (default): consecutive steps overlap under the phase gate - this step's cleanup chain (one workgroup\n"
                         "per image) under the next step's pyramid, its descriptors under the next extrema / gradient pass.\n"
                         "1: one step at a time")
    ap.add_argument("--rccl-loopback", action="store_true",
                    help="N = 1 only: every step's keypoint lists also travel through RCCL point-to-point to this same rank "
                         "(KeypointGather(loopback=True)): the N > 1 gather path, messages and sizes, on a one-GPU box")
    ap.add_argument("--repeats", type=int, default=5, help="repetitions of the K-step loop reported as ms_per_step_repeats (the headline is the first)")
    ap.add_argument("--check-gather", action="store_true", default=True,
                    help="N > 1, after the timed region (the default since round 5: a multi-GPU line carries its own evidence): rank 0 runs every "
                         "rank's frames itself, rank by rank, and compares the lists that arrived through the gather in the last step with its "
                         "own - records, descriptor floats and per-image counts in global image order (seeds 1 .. N x frames); the line then "
                         "carries gather_check")
    ap.add_argument("--no-check-gather", dest="check_gather", action="store_false", help="skip that check (it costs N batches on rank 0 after the timed region)")
    ap.add_argument("--pipeline-gate", type=int, default=1, choices=[0, 1],
                    help="pipeline depth > 1: 1 (default) joins the contexts with a phase gate (sift_amd/csrc/phase_gate.h); 0 leaves the interleaving to the GPU's queues")
    args = ap.parse_args()

    global W, H, DOGS, OCTAVES, SUBPIXEL
    W, H, DOGS, OCTAVES, SUBPIXEL, frames_default, label = WORKLOADS[args.workload]
    if args.frames is None:
        args.frames = frames_default
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: this process becomes the launcher.  It has not touched the GPU (no torch import,
        # no HIP call so far) and never will: N fresh children, one rank per GPU, rank 0's JSON line passed through.
        raise SystemExit(launch_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         f"or bare (no WORLD_SIZE in the environment) to let bench.py start its own ranks")

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
    loopback = bool(args.rccl_loopback) and world == 1
    rccl_ranks = 0
    if loopback:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:   # a free port of this host (a fixed one can still be held by the run before)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    if world > 1 or loopback:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)                       # creates the RCCL communicator; every rank contributes 1
            rccl_ranks = int(probe.item()) if int(probe.item()) == dist.get_world_size() else -1
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
    options = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in args.set)
    if (world > 1 or loopback) and args.wire == "sparse":
        options.setdefault("wire_count", 1)   # the descriptor kernel also counts what the sparse wire format will carry
    pipe = BatchPipeline(local_rank, depth, options, gated=bool(args.pipeline_gate))
    ctxs = pipe.contexts
    ctx = ctxs[0]
    params = _lib.Params(DOGS, OCTAVES, SIGMA, K_SQRT2, SUBPIXEL)
    L = ctx._L

    from sift_amd.gather import GatherThread, KeypointGather, device_results

    # N > 1: the gather of step k (keypoint records + descriptors to rank 0 over RCCL point-to-point, never images) rides
    # behind the header of step k+1 and overlaps the kernels of the following steps, which run on the library's own
    # streams (sift_amd/gather.py: KeypointGather, no per-step collective); every transfer completes inside the timed region.
    gatherer = KeypointGather(nf, comm_dev, dst=0, loopback=loopback, concat=False) if (world > 1 or loopback) else None
    gathered = [0, 0]       # steps and keypoints that have arrived on rank 0
    keep = []               # tensors of the last pushes (the gather reads them until two pushes later)
    tickets = []            # submitted steps whose results have not been collected yet (at most depth - 1 between steps)

    last_gathered = [None]  # the newest step that arrived on rank 0: (records_all, values_all, counts_all)

    def note(done):
        for recs, vals, counts in done:
            gathered[0] += 1
            gathered[1] += int(counts.sum())
            last_gathered[0] = (recs, vals, counts)

    def collect(ticket):
        """Finish one step: wait for its batch, hand its keypoint lists to the gather (N > 1), free its slot."""
        c = ticket.result()
        total = c.total()
        if gatherer is not None:
            # wire format (lossless, sift_amd/gather.py): sparse = 34-byte records (20 + 112 presence bits) + the floats that are set
            rec_out = None
            if args.wire == "sparse" and ((comm_dev.type != "cpu" and gatherer.rank != gatherer.dst) or loopback):
                rec_out = gatherer.records_buffer(total * 34)   # the records are packed straight into their message buffer
            kp, desc = device_results(c, total, dev, wire=args.wire, rec_out=rec_out)
            counts = c.counts()
            if comm_dev.type == "cpu":     # test backend: stage through host memory
                kp, desc = kp.cpu(), desc.cpu()
            keep.append((kp, desc))
            del keep[:-3]
            note(gatherer.push(kp, desc, counts))
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
        return total

    slot_streams = [torch.cuda.Stream(device=dev) for _ in range(depth)] if gatherer is not None else []

    def run_steps(n_steps):
        """`n_steps` passes over the batch through the host loop the run was asked for; returns the keypoints they produced.
        The warm-up takes the same path as the timed region (same threads, torch streams and allocator pools)."""
        if not (args.host_loop == "stream" and depth > 1):
            kps_ = 0
            for _ in range(n_steps):
                kps_ += step()
            kps_ += drain()
            if gatherer is not None:
                note(gatherer.flush())
            return kps_
        # every context's host thread takes its next step itself (BatchPipeline.run_stream): no dispatching thread between the end
        # of a context's batch and the start of its next.  N > 1: that thread also packs the batch's lists into the wire format
        # (beside the other context's batch) and hands them to ONE gather thread, which pushes them in step order - the
        # point-to-point messages of a rank are then issued by one thread, in the same order on every rank.
        import threading
        left, per_slot = [n_steps], [0] * depth
        item = (d_frames.data_ptr(), nf, W, H, params)
        seq_next, seq_of = [0], {}

        def source():
            if left[0] <= 0:
                return None
            left[0] -= 1
            seq_of[threading.get_ident()] = seq_next[0]     # (called under run_stream's lock, on the thread that will run the batch)
            seq_next[0] += 1
            return item

        # A torch stream per context thread: the message buffers are allocated on it and nothing else ever runs on it, so the
        # thread never waits for the gather's transfers (torch's default stream, the gather thread's, is made to wait for every
        # receive it posts; allocating there and synchronising it - what device_results does before it hands memory to the
        # library's kernels - would park this thread behind the previous steps' messages).  (slot_streams, made once: the
        # allocator keeps a pool per stream, and the warm-up is there to fill them.)
        def sink(c, slot, _item):
            total = c.total()
            per_slot[slot] += total
            if gatherer is None:
                return
            torch.cuda.set_device(local_rank)
            with torch.cuda.stream(slot_streams[slot]):
                rec_out = None
                if args.wire == "sparse" and ((comm_dev.type != "cpu" and gatherer.rank != gatherer.dst) or loopback):
                    rec_out = gatherer.records_buffer(total * 34)   # the records are packed straight into their message buffer
                # the pack is only queued (side stream): this thread goes straight on to the context's next batch, whose descriptor
                # stage waits for it on the device; the gather thread waits for it on the host before it pushes the lists
                defer = args.wire == "sparse" and comm_dev.type != "cpu" and total > 0
                kp, desc = device_results(c, total, dev, wire=args.wire, rec_out=rec_out, defer_pack=defer)
            counts = c.counts().copy()
            if comm_dev.type == "cpu":     # test backend: stage through host memory
                kp, desc = kp.cpu(), desc.cpu()
            gthread.put(seq_of[threading.get_ident()], kp, desc, counts, ready=c.pack_wait if defer else None)

        gthread = GatherThread(gatherer, note, cuda_device=local_rank) if gatherer is not None else None
        try:
            pipe.run_stream(source, sink)
        finally:
            if gthread is not None:
                gthread.close()      # pushes what is queued, flushes the gather, re-raises what the thread raised
        return sum(per_slot)

    # N > 1: the gather path reaches its steady state only after a dozen steps (the transport's first messages set up its
    # connections, lists arrive two pushes after they went in, and torch's allocator has to have seen every message buffer of the
    # cycle once: measured on one GPU through RCCL, 4.8 ms per step over the 20 steps behind a warm-up of 4, 2.98 behind one of 20).
    # That is set-up, not a step of the workload: it is done before the W warm-up steps the caller asked for, untimed like them.
    # N = 1: a context's first two batches run alone in the process (the library serialises the runtime's lazy first-use set-up:
    # context.cpp, FirstBatch), so the first 2 x depth steps are set-up as well, whatever W the caller passes.
    settle = GATHER_SETTLE_STEPS if gatherer is not None else 2 * depth
    if settle:
        run_steps(settle)
        if gatherer is not None:
            gatherer = KeypointGather(nf, comm_dev, dst=0, loopback=loopback, concat=False)
    if args.warmup:
        run_steps(args.warmup)
    if gatherer is not None:     # the warm-up steps' lists were gathered too, before the clock starts
        gatherer = KeypointGather(nf, comm_dev, dst=0, loopback=loopback, concat=False)
        gatherer_t0 = (gathered[0], gathered[1])
    for c in ctxs:
        c.set_option("profile", args.profile_every)   # every N-th batch of a context carries the per-launch timing events
        c.profile_reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    kps = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    for c in ctxs:
        c.set_option("profile", 0)
    gathered_head = (gathered[0] - gatherer_t0[0], gathered[1] - gatherer_t0[1]) if gatherer is not None else (None, None)
    # the same K steps a few more times, AFTER the headline region (which stays the first repetition, so that `steps` and
    # `ms_per_step` describe one and the same loop): the spread of a 55 ms sample on this box, max over ranks each
    repeat_ms = [dt / args.steps * 1e3]
    for _ in range(max(0, args.repeats - 1)):
        if gatherer is not None:
            gatherer = KeypointGather(nf, comm_dev, dst=0, loopback=loopback, concat=False)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t_r = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        repeat_ms.append((time.perf_counter() - t_r) / args.steps * 1e3)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        k = torch.tensor([kps], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
        kps = int(k.item())
        rr = torch.tensor(repeat_ms, dtype=torch.float64, device=comm_dev)
        dist.all_reduce(rr, op=dist.ReduceOp.MAX)
        repeat_ms = [float(v) for v in rr.tolist()]

    if rank == 0:
        prof = [c.profile(0) for c in ctxs]
        ms, launches, nbytes = (sum(p[i] for p in prof) for i in range(3))
        # The top Gaussian level of an octave runs on the side stream beside the next octave's first launches: launches overlap, so the family's rate is its bytes over the time during which at least one
        # blur launch was running (union of the launches' [start, stop] intervals, from the same events), not over the sum of
        # the durations, which counts the overlapped time twice.
        busy_ms = sum(c.profile_busy_ms(0) for c in ctxs)
        achieved = (nbytes / 1e9) / (busy_ms / 1e3) if busy_ms > 0 else 0.0
        achieved_sum = (nbytes / 1e9) / (ms / 1e3) if ms > 0 else 0.0
        out = {
            "metric": "keypoints/sec, 1920x1080 4oct/3DoG" if args.workload == "config4" else f"keypoints/sec, {args.workload} (not the headline metric)",
            "value": kps / dt,
            "unit": "keypoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_repeats": {"n": len(repeat_ms), "min": min(repeat_ms), "median": sorted(repeat_ms)[len(repeat_ms) // 2], "max": max(repeat_ms),
                                    "all": repeat_ms, "what": "the same K-step loop repeated back to back; the headline ms_per_step is the first repetition"},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": label.format(n=nf),
                       "frames_per_gpu": nf, "frames_total": nf * world, "pipeline_depth": depth, "host_loop": (args.host_loop if depth > 1 else "dispatch"), "pipeline_gate": bool(args.pipeline_gate) and depth > 1, "keypoints_per_step": kps // max(args.steps, 1),
                       "frames_per_s": nf * world * args.steps / dt,
                       "rccl_ranks": rccl_ranks,   # as the RCCL communicator reports it (0: no RCCL communicator in this run)
                       "rccl_loopback": loopback,   # N = 1 with the N > 1 gather messages sent through RCCL to this same rank
                       "setup_steps": settle,   # untimed set-up steps in front of the W warm-up steps: the library's first batches (2 per context); N > 1: the gather path settling (16)
                       "gather_steps_on_rank0": gathered_head[0],
                       "gather_keypoints_on_rank0": gathered_head[1],
                       "gather_ms_per_step": (gatherer.wait_s / args.steps * 1e3) if gatherer is not None else None,
                       "wire_bytes_per_step": (gatherer.wire_bytes / args.steps) if gatherer is not None else None,
                       "gather": ("RCCL p2p of keypoint records + descriptors to rank 0, no per-step collective (sizes ride one step ahead), overlapped with the following steps; "
                                  + {"full": "128 floats per descriptor", "packed": "descriptors on the wire as 112 of 128 floats (bin 7 of each cell is structurally +0.0f; lossless)",
                                     "sparse": "descriptors on the wire as 112 presence bits + the floats that are not +0.0f (about a third; lossless)"}[args.wire]) if gatherer is not None else "none (1 GPU)"},
            "roofline": {"kernel": "blur_pair_kernel / blur_stream2_kernel / blur_reduce_kernel / blur_fused_kernel (separable Gaussian: every launch of the pyramid)",
                         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                         "timing": "hipEventElapsedTime over start/stop events attached to each blur dispatch (hipExtLaunchKernelGGL) on the library's streams, inside the timed region; "
                                   f"every {args.profile_every}th batch of a context is instrumented (the events keep consecutive launches ~10 us apart); "
                                   "achieved = algorithmic bytes of all blur launches / the time during which at least one of them was running (union of the "
                                   "launches' intervals: an octave's top level overlaps the next octave's first launches on a second stream); in the timed region "
                                   "the previous step's cleanup chain runs beside the pyramid (see frac_alone for the launches alone on the chip)",
                         "launches": launches, "avg_launch_ms": busy_ms / launches if launches else None,
                         "algorithmic_bytes_per_launch": nbytes / launches if launches else None,
                         # the same bytes over the SUM of the launches' durations (overlapped time counted twice)
                         "achieved_over_sum_of_durations": achieved_sum, "frac_over_sum_of_durations": achieved_sum / HBM_PEAK_GBS,
                         "sum_of_durations_ms_per_launch": ms / launches if launches else None},
        }
        # The three kernels of the OTHER phase of a step (E || D: the extremum scan and the gradient maps of one batch beside the
        # partner batch's descriptors), from the same kind of events on their own dispatch packets: strict (bytes over the sum of
        # the launches' durations) and union (over the time at least one launch of the class was running) in the timed region;
        # `alone` is added below from the steps run one at a time.  Algorithmic bytes: SURVEY.md section 8(d).
        batches_timed = sum(c.profile_batches() for c in ctxs)
        kp_per_batch = kps / max(args.steps, 1) / max(world, 1)
        for key, which, kernel in (("descriptor", 2, "descriptor_wave_kernel"), ("extrema", 3, "extrema_edge_kernel"), ("gradient", 4, "gradient4_kernel")):
            pk = [c.profile(which) for c in ctxs]
            k_ms, k_n, k_b = (sum(q[i] for q in pk) for i in range(3))
            k_busy = sum(c.profile_busy_ms(which) for c in ctxs)
            if not k_n or k_ms <= 0:
                continue
            if which == 2:
                k_b = 3584.0 * kp_per_batch * batches_timed     # 3.5 KB per keypoint, the keypoints of the instrumented batches
            out["roofline"][key] = {"kernel": kernel, "launches": k_n, "algorithmic_bytes_per_launch": k_b / k_n,
                                    "avg_launch_ms": k_ms / k_n,
                                    "frac_over_sum_of_durations": k_b / 1e9 / (k_ms / 1e3) / HBM_PEAK_GBS,
                                    "frac": k_b / 1e9 / (k_busy / 1e3) / HBM_PEAK_GBS if k_busy > 0 else None}
        # The same launches against the OTHER roof: from radius 7 on a blur is bound by instruction issue, not by HBM (DESIGN.md
        # section 7, round 5) - the family's lane-operations over the same busy time, against the chip's FP32 vector rate
        if not SUBPIXEL and busy_ms > 0 and launches:
            # (the number of instrumented batches comes from the library - sift_hip_profile_batches -, not from an assumed
            # number of launches per batch: a batch whose first two levels do not take the pair launch makes 16, not 15)
            if batches_timed and args.workload == "config4":
                lane_ops = blur_lane_ops(nf, W, H, OCTAVES, DOGS, SIGMA, K_SQRT2) * batches_timed
                out["roofline"]["valu"] = {"achieved": lane_ops / 1e12 / (busy_ms / 1e3), "peak": VALU_PEAK_TLANEOPS, "unit": "T lane-op/s",
                                           "frac": lane_ops / 1e12 / (busy_ms / 1e3) / VALU_PEAK_TLANEOPS,
                                           "issue_ceiling_at_2_waves_per_simd": VALU_ISSUE_CEILING_2_WAVES,
                                           "frac_of_issue_ceiling": lane_ops / 1e12 / (busy_ms / 1e3) / VALU_ISSUE_CEILING_2_WAVES,
                                           "lane_ops_per_batch": lane_ops / batches_timed,
                                           "what": "FP32 multiplies + adds of the blur launches (7R+4 per pixel and level, the reference's unfused arithmetic) over the "
                                                   "same busy time, against the data sheet's 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz and against what packed multiply / add "
                                                   "streams were measured to issue at two waves per SIMD (tools/probe/valu_rate_probe.hip); bookkeeping instructions not counted"}
        # the whole step against the same peak: every stage's algorithmic bytes over the step's time
        if args.workload == "config4" and not SUBPIXEL:
            out["roofline"]["whole_step"] = whole_step_fraction(nf, W, H, kps / max(args.steps, 1) / max(world, 1), dt / args.steps)
        if args.check_gather and gatherer is not None and not loopback:
            # ---- what arrived through the gather against this rank's own run of EVERY rank's frames (outside the timed region)
            recs_parts, vals_parts, counts_all = last_gathered[0]     # one tensor per rank (KeypointGather(concat=False)), rank order
            ok, io = len(recs_parts) == world and len(vals_parts) == world, 0
            for r in range(world):
                fr = np.stack([synth_frame(W, H, r * nf + i + 1) for i in range(nf)])
                ctx.calculate_batch(fr, params)
                tot = ctx.total()
                kp_r, val_r = device_results(ctx, tot, dev, wire=args.wire)
                kp_r, val_r, cnt_r = kp_r.cpu(), val_r.cpu(), ctx.counts().copy()
                ok = ok and bool((counts_all[io:io + nf].cpu().numpy() == cnt_r).all())
                ok = ok and recs_parts[r].cpu().numpy().tobytes() == kp_r.numpy().tobytes()
                ok = ok and vals_parts[r].cpu().numpy().tobytes() == val_r.numpy().tobytes()
                io += nf
            ok = ok and io == counts_all.numel()
            out["config"]["gather_check"] = bool(ok)
            out["config"]["gather_check_what"] = (f"the last step's lists on rank 0 (records, descriptor floats, counts of {world * nf} images) equal "
                                                  f"rank 0's own run of every rank's frames, seeds 1..{world * nf} in order")
        if world == 1:
            # ---- the blur family ALONE on the chip: a few steps one at a time after the timed region, every one instrumented
            ctx.set_option("profile", 1)
            ctx.profile_reset()
            for _ in range(3):
                ctx.calculate_batch_device(d_frames.data_ptr(), nf, W, H, params)
            ctx.set_option("profile", 0)
            ms_a, launches_a, nbytes_a = ctx.profile(0)
            busy_a = ctx.profile_busy_ms(0)
            for key, which in (("descriptor", 2), ("extrema", 3), ("gradient", 4)):
                k_ms, k_n, k_b = ctx.profile(which)
                if key in out["roofline"] and k_n and k_ms > 0:
                    if which == 2:
                        k_b = 3584.0 * kp_per_batch * ctx.profile_batches()
                    out["roofline"][key]["avg_launch_ms_alone"] = k_ms / k_n
                    out["roofline"][key]["frac_alone"] = k_b / 1e9 / (k_ms / 1e3) / HBM_PEAK_GBS
            if busy_a > 0:
                out["roofline"]["achieved_alone"] = (nbytes_a / 1e9) / (busy_a / 1e3)
                out["roofline"]["frac_alone"] = out["roofline"]["achieved_alone"] / HBM_PEAK_GBS
                out["roofline"]["alone_what"] = ("the same events and bytes over 3 steps run one at a time AFTER the timed region: the blur launches alone on the chip "
                                                 "(in the timed region the previous step's cleanup chain shares the chip with them, phase gate schedule 1)")
        if world == 1 and not args.no_extras:
            # ---- the boundary as a host caller sees it (main.cpp:56-57 hands over host memory and reads the vector back):
            # frames from host memory in, keypoints + descriptors to host memory out, two batches in flight
            from sift_amd.sift import pinned_array

            # three batches in flight for this leg (upload, kernels and download of different batches side by side): a pipeline of its own
            hdepth = 3
            hpipe = BatchPipeline(local_rank, hdepth, options, gated=bool(args.pipeline_gate))

            def host_loop(src, fetch, n_steps, on_worker=False):
                """n_steps batches from host memory `src`, `hdepth` in flight; `fetch(context, slot)` brings the oldest batch's lists to the host -
                on the dispatching thread, or (on_worker) on the batch's own worker thread right behind the batch, beside the other batches.
                (One dispatching thread: this loop is bound by the link and the copies' interference with the kernels, and letting every
                context's thread feed itself - BatchPipeline.run_stream, the headline loop - made no difference here: 8.0 against 7.4 - 7.6 ms
                on the box of that comparison.)"""
                pend, total, got = [], 0, {}
                t_0 = time.perf_counter()
                for i in range(n_steps + hdepth):
                    if i < n_steps:
                        if on_worker:
                            pend.append(hpipe.submit(src, params, then=lambda c, slot: got.__setitem__(slot, fetch(c, slot))))
                        else:
                            pend.append(hpipe.submit(src, params))
                    if pend and (len(pend) >= hdepth or i >= n_steps):
                        tk = pend.pop(0)
                        c = tk.result()
                        total += got.pop(tk.slot) if on_worker else fetch(c, tk.slot)
                        tk.release()
                return (time.perf_counter() - t_0) / n_steps, total // n_steps

            from sift_amd.sift import unpack_sparse_host
            cap = int(out["config"]["keypoints_per_step"] * 1.25) + 1024
            hs = max(2, min(args.steps, 20))
            # (a) what a host that reads 8-bit files holds (the reference's inputs are 8-bit files, main.cpp:52-54): uint8 frames in
            # (widened on the GPU), keypoint lists out in the sparse wire format (34-byte records + the descriptor floats that are set)
            frames_u8 = frames.astype(np.uint8)
            assert np.array_equal(frames_u8.astype(np.float32), frames)
            pin_u8 = pinned_array(frames_u8.shape, np.uint8)
            pin_u8[...] = frames_u8
            pin_sp = [(pinned_array((cap, 34), np.uint8), pinned_array((cap * 64,), np.float32)) for _ in range(hdepth)]
            wire = [0]
            for c in hpipe.contexts:
                c.set_option("wire_count", 1)     # the descriptor kernel counts the floats the sparse format will carry

            def fetch_sparse(c, slot):
                rec, val = c.results_sparse(pin_sp[slot][0], pin_sp[slot][1])
                wire[0] = rec.nbytes + val.nbytes
                return rec.shape[0]

            dense = [(np.empty(cap, _lib.KEYPOINT_DTYPE), np.empty((cap, 128), np.float32)) for _ in range(hdepth)]

            def fetch_sparse_unpacked(c, slot):   # ... and expanded again on the host into records + 128-float descriptors
                rec, val = c.results_sparse(pin_sp[slot][0], pin_sp[slot][1])
                unpack_sparse_host(rec, val, dense[slot][0], dense[slot][1], threads=min(16, os.cpu_count() or 1))
                return rec.shape[0]

            host_loop(pin_u8, fetch_sparse, 3 * hdepth)     # set-up: every context's first batches run alone (FirstBatch), buffers grow
            t_sp, k_sp = host_loop(pin_u8, fetch_sparse, hs)
            host_loop(pin_u8, fetch_sparse_unpacked, hdepth, on_worker=True)
            t_spu, _ = host_loop(pin_u8, fetch_sparse_unpacked, hs, on_worker=True)
            for c in hpipe.contexts:
                c.set_option("wire_count", int(options.get("wire_count", 0)))
            # (b) float32 frames in, dense 532-byte records out (round 2's figure)
            pin_frames = pinned_array(frames.shape, np.float32)
            pin_frames[...] = frames
            pin_out = [(pinned_array((cap,), _lib.KEYPOINT_DTYPE), pinned_array((cap, 128), np.float32)) for _ in range(hdepth)]
            fetch_dense = lambda c, slot: c.results(pin_out[slot][0], pin_out[slot][1])[0].size   # noqa: E731
            host_loop(pin_frames, fetch_dense, hdepth)
            t_pin, k_pin = host_loop(pin_frames, fetch_dense, hs)
            fetch_page = lambda c, slot: c.results()[0].size   # noqa: E731
            host_loop(frames, fetch_page, 1)
            t_page, _ = host_loop(frames, fetch_page, max(2, hs // 4))
            nbytes_io = frames.nbytes + k_pin * (20 + 512)
            nbytes_sp = frames_u8.nbytes + wire[0]
            out["host_inclusive"] = {"ms_per_step": t_sp * 1e3, "keypoints_per_s": k_sp / t_sp,
                                     "pcie_gbytes_per_step": nbytes_sp / 1e9, "pcie_gb_per_s": nbytes_sp / 1e9 / t_sp,
                                     "what": "8-bit frames from page-locked host memory in (sift_hip_calculate_batch_u8: a quarter of the bytes, widened on the GPU to the "
                                             "floats vigra::importImage yields), keypoint lists to page-locked host memory out in the lossless sparse format "
                                             "(sift_hip_result_copy_sparse: 34-byte records + the descriptor floats that are not +0.0f), per step; three batches in flight (BatchPipeline depth 3)",
                                     "with_host_unpack_ms_per_step": t_spu * 1e3,
                                     "with_host_unpack_what": "the same plus sift_hip_sparse_unpack_host (AVX2) on up to 16 host threads, run by the batch's own worker thread right behind its download (`then=`), beside the other batches: dense 20-byte records + 128-float descriptors in ordinary memory",
                                     "float_dense_ms_per_step": t_pin * 1e3, "float_dense_keypoints_per_s": k_pin / t_pin,
                                     "float_dense_pcie_gbytes_per_step": nbytes_io / 1e9, "float_dense_pcie_gb_per_s": nbytes_io / 1e9 / t_pin,
                                     "float_dense_what": "float32 frames in, 20-byte records + 128-float descriptors out, page-locked memory on both sides (round 2's boundary)",
                                     "pageable_ms_per_step": t_page * 1e3,
                                     "pageable_what": "float32 / dense with ordinary (pageable) numpy arrays on both sides: chunked through the library's pinned staging buffers"}
            hpipe.close()
            one = pinned_array((1,) + frames.shape[1:], np.float32)
            one[...] = frames[:1]
            lat = []
            for _ in range(12):
                t_0 = time.perf_counter()
                ctx.calculate_batch(one, params)
                ctx.results(pin_out[0][0], pin_out[0][1])
                lat.append(time.perf_counter() - t_0)
            out["single_frame_ms"] = float(np.median(lat[2:]) * 1e3)
            out["dropin_cpp"] = dropin_cpp_leg(frames[0])
            if args.pmc_traffic:
                extra = ["--workload", args.workload, "--frames", str(nf)] + [a for kv in args.set for a in ("--set", kv)]
                traffic, src = pmc_traffic_live(extra)
                out["roofline"]["traffic"] = traffic
                out["roofline"]["traffic_source"] = src
        if world == 1 and not args.no_cpu_baseline and args.workload == "config4":
            out["cpu_baseline"] = cpu_baseline(frames[:CPU_SAMPLE_FRAMES])
            # ---- result check of this very run, outside the timed region: one more step of the batch through the same pipeline,
            # its first CPU_SAMPLE_FRAMES images' keypoints and descriptors against what the oracle just computed for those frames
            per_frame = out["cpu_baseline"].pop("per_frame")
            tk = pipe.submit_device(d_frames.data_ptr(), nf, W, H, params)
            c = tk.result()
            cnt = c.counts().copy()
            kp_h, de_h = c.results()
            ok, base = True, 0
            for i, (n_want, digest) in enumerate(per_frame):
                n = int(cnt[i])
                ok = ok and n == n_want and _result_digest(kp_h[base:base + n], de_h[base:base + n]) == digest
                base += n
            tk.release()
            out["parity_spot_check"] = bool(ok)
            out["parity_spot_check_what"] = (f"frames 1..{len(per_frame)} of one more step of this batch: per-image keypoint count and SHA-256 over "
                                            "(x, y, octave, index, scale, orientation, 128-float descriptors) equal the oracle's")
        os.write(_JSON_FD, (json.dumps(out) + "\n").encode())
    pipe.close()
    if world > 1 or loopback:
        dist.destroy_process_group()


# The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints its version banner to stdout when a communicator is
# created), so the line goes to a private copy of the original stdout and file descriptor 1 is pointed at stderr for everything else.
_JSON_FD = 1

if __name__ == "__main__":
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    main()
