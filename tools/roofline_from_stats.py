#!/usr/bin/env python3
"""Roofline of the Gaussian-pyramid kernels from rocprofv3 output, recomputed from first principles.

    python tools/roofline_from_stats.py --stats profiles/r02_kernel_stats.csv                    family total
    python tools/roofline_from_stats.py --trace gpurun_out/prof/.../kernel_trace.csv [--dump profiles/r02_blur_launches.csv]
    python tools/roofline_from_stats.py --launches profiles/r02_blur_launches.csv                per-launch table

The plan (which blur runs on which level, with which radius) is rebuilt here from the workload's parameters exactly as
Sift::_createDOGs schedules it (/root/reference/sift.cpp:388-411; radius = (int)(3 sigma + 0.5), Vigra initGaussian), and
every launch is priced at its ALGORITHMIC bytes:
    g(0,0)                        4 B read + 4 B written per pixel                      8 N
    level blur (round 5)          4 B read + 4 B written: no DoG level is written any more   8 N
                                  (option dog_in_extrema, the default; --dog-in-extrema 0 prices the old plan:)
    level blur + DoG              4 B read + 4 B + 4 B written                          12 N
    top level of an octave        4 B read + 4 B written: the DoG only (option lazy_top,  8 N
                                  the default since round 4; --lazy-top 0 prices the old 12 N)
    reduceToNextLevel, fused      4 B read per source pixel + 4 B per KEPT pixel         4 N + 4 N_next
    reduceToNextLevel, unfused    the blur alone (the resampling launch is not a blur)   8 N
against the HBM3E peak of 8 TB/s (MI355X_MICROARCH.md).  `--trace` walks the dispatches in order and matches the i-th blur
launch of a batch step to the i-th blur of the plan (the template radius must agree); `--stats` only needs the per-kernel
totals and gives the family figure bench.py reports (sum of algorithmic bytes / sum of durations)."""
import argparse
import csv
import math
import re
import sys

import numpy as np

PEAK = 8000.0  # GB/s


def plan(w, h, n, dogs, octaves, sigma, k, subpixel, lazy_top=True, dog_in_extrema=True, blur_pair=True):
    """[(what, octave, level, radius, pixels per launch, algorithmic bytes if fused, ... unfused)]"""
    ops = []
    f32 = np.float32
    if subpixel:
        ops.append(("increase pre-blur", 0, 0, max(1, int(3.0 * 1.0 + 0.5)), w * h * n, 8.0 * w * h * n, 8.0 * w * h * n))
        w, h = 2 * w, 2 * h
    ws, hs = [w], [h]
    for _ in range(1, octaves):
        ws.append((ws[-1] + 1) // 2)
        hs.append((hs[-1] + 1) // 2)
    gs = {(0, 0): f32(sigma)}
    exp = 0
    for i in range(octaves):
        for j in range(1, dogs + 1):
            gs[(i, j)] = f32(math.pow(float(f32(k)), float(exp)) * float(f32(sigma)))
            exp += 1
        if i < octaves - 1:
            gs[(i + 1, 0)] = gs[(i, dogs - 1)]
            exp -= 2
    rad = lambda s: max(1, int(3.0 * float(s) + 0.5))  # noqa: E731
    # option blur_pair (with dog_in_extrema): the first two levels are one launch, one read and two levels written (kernels_pair.hip)
    pair = blur_pair and dog_in_extrema and rad(gs[(0, 0)]) == rad(gs[(0, 1)]) and 3 <= rad(gs[(0, 0)]) <= 6
    if pair:
        ops.append(("pair: g(0,0) and g(0,1)", 0, 0, rad(gs[(0, 0)]), ws[0] * hs[0] * n, 12.0 * ws[0] * hs[0] * n, 12.0 * ws[0] * hs[0] * n))
    else:
        ops.append(("g(0,0)", 0, 0, rad(gs[(0, 0)]), ws[0] * hs[0] * n, 8.0 * ws[0] * hs[0] * n, 8.0 * ws[0] * hs[0] * n))
    for o in range(octaves):
        px = ws[o] * hs[o] * n
        for j in range(1, dogs + 1):
            if pair and o == 0 and j == 1:
                continue
            if dog_in_extrema:   # round 5's default: the pyramid writes Gaussian levels only, every level of them (context.cpp: dog_in_extrema)
                ops.append((f"g({o},{j})", o, j, rad(gs[(o, j)]), px, 8.0 * px, 8.0 * px))
            elif j == dogs and lazy_top:   # the top Gaussian level only feeds this DoG and is not written (context.cpp: lazy_top)
                ops.append((f"dog({o},{j - 1}) [g({o},{j}) not kept]", o, j, rad(gs[(o, j)]), px, 8.0 * px, 8.0 * px))
            else:
                ops.append((f"g({o},{j}) + dog({o},{j - 1})", o, j, rad(gs[(o, j)]), px, 12.0 * px, 12.0 * px))
        if o < octaves - 1:
            pd = ws[o + 1] * hs[o + 1] * n
            ops.append((f"reduce g({o},{dogs - 1}) -> g({o + 1},0)", o, dogs - 1, rad(gs[(o, dogs - 1)]), px, 4.0 * px + 4.0 * pd, 8.0 * px))
    return ops


def parse_name(name):
    m = re.search(r"blur_reduce_kernel<(\d+), (\d+)>", name)   # reduceToNextLevel, kept pixels only (kernels_reduce.hip)
    if m:
        return {"form": "reduce", "R": int(m.group(1)), "dog": False, "dec": True}
    m = re.search(r"blur_pair_kernel<(\d+)>", name)   # the first two levels in one launch (kernels_pair.hip)
    if m:
        return {"form": "pair", "R": int(m.group(1)), "dog": False, "dec": False}
    m = re.search(r"blur_stream2_kernel<(\d+), (\d+)>", name)   # round 6: rows packed in pairs, a Gaussian level out (blur_stream.h)
    if m:
        return {"form": "stream", "R": int(m.group(1)), "dog": False, "dec": False}
    m = re.search(r"blur_(stream|fused)_kernel<(\d+), (true|false)(?:, (\d+))?(?:, (true|false))?>", name)
    if not m:
        return None
    return {"form": m.group(1), "R": int(m.group(2)), "dog": m.group(3) == "true", "dec": m.group(5) == "true"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--trace")
    ap.add_argument("--launches")
    ap.add_argument("--dump", help="with --trace: write the blur launches (name, start, end) to this small csv")
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--lazy-top", type=int, default=1, choices=[0, 1])
    ap.add_argument("--dog-in-extrema", type=int, default=1, choices=[0, 1], help="1 (round 5's default): no blur launch writes a DoG level, 8 B per pixel each")
    ap.add_argument("--blur-pair", type=int, default=1, choices=[0, 1], help="1 (the default build): g(0,0) and g(0,1) are one launch, 12 B per pixel")
    ap.add_argument("--dogs", type=int, default=3)
    ap.add_argument("--octaves", type=int, default=4)
    ap.add_argument("--sigma", type=float, default=1.6)
    ap.add_argument("--k", type=float, default=float(np.float32(np.sqrt(2.0))))
    ap.add_argument("--subpixel", type=int, default=0)
    a = ap.parse_args()
    ops = plan(a.w, a.h, a.n, a.dogs, a.octaves, a.sigma, a.k, a.subpixel, bool(a.lazy_top), bool(a.dog_in_extrema), bool(a.blur_pair))
    per_step_bytes = None

    if a.stats:
        tot_ns, calls = 0.0, 0
        for r in csv.DictReader(open(a.stats)):
            if parse_name(r["Name"]):
                tot_ns += float(r["TotalDurationNs"])
                calls += int(r["Calls"])
        if calls % len(ops):
            print(f"warning: {calls} blur launches is not a multiple of the plan's {len(ops)}", file=sys.stderr)
        steps = calls / len(ops)
        # which reduce launches ran fused is visible from the names: a decimating stream kernel exists or not
        names = [r["Name"] for r in csv.DictReader(open(a.stats))]
        dec_R = {parse_name(nm)["R"] for nm in names if parse_name(nm) and parse_name(nm)["dec"]}
        per_step_bytes = sum(op[5] if (not op[0].startswith("reduce") or op[3] in dec_R) else op[6] for op in ops)
        gbs = per_step_bytes * steps / tot_ns
        print(f"blur family: {calls} launches over {steps:g} steps, {tot_ns / calls / 1e3:.2f} us and {per_step_bytes / len(ops) / 1e6:.2f} MB per launch "
              f"-> {gbs:.0f} GB/s = {gbs / PEAK:.3f} of {PEAK:.0f} GB/s")

    rows = []
    if a.trace:
        for r in csv.DictReader(open(a.trace)):
            if parse_name(r["Kernel_Name"]):
                rows.append((r["Kernel_Name"].split("(")[0].replace("void sift_hip::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        rows.sort(key=lambda t: t[1])
        if a.dump:
            with open(a.dump, "w", newline="") as f:
                wr = csv.writer(f)
                wr.writerow(["kernel", "start_ns", "end_ns"])
                wr.writerows(rows)
    elif a.launches:
        rows = [(r["kernel"], int(r["start_ns"]), int(r["end_ns"])) for r in csv.DictReader(open(a.launches))]
    if rows:
        # Two contexts interleave their pyramids only at whole-pyramid granularity (the phase gate), so the launches come in
        # runs of len(ops).  Inside a run the order is not the plan's: an octave's top level runs on the side stream beside the
        # next octave's first launches (option pyramid_side), so a launch is matched to the plan by (radius, DoG, decimating)
        # and, among equal signatures, by octave order (the dependency chain keeps those in start order).
        if len(rows) % len(ops):
            print(f"warning: {len(rows)} blur launches is not a multiple of the plan's {len(ops)}", file=sys.stderr)
        acc = [[0.0, 0, 0.0] for _ in ops]
        spans, busy = [], []
        for g0 in range(0, len(rows) - len(ops) + 1, len(ops)):
            grp = rows[g0:g0 + len(ops)]
            free = list(range(len(ops)))
            for name, t0, t1 in grp:
                p = parse_name(name)
                def fits(op):
                    has_dog, is_reduce = "dog(" in op[0], op[0].startswith("reduce")
                    if (p["form"] == "pair") != op[0].startswith("pair"):
                        return False
                    return op[3] == p["R"] and ((p["dog"] and has_dog) or (p["dec"] and is_reduce) or (not p["dog"] and not p["dec"] and not has_dog))
                k = next((i for i in free if fits(ops[i])), None)
                assert k is not None, f"{name} matches no launch of the plan that is still open in its pyramid"
                free.remove(k)
                op = ops[k]
                nbytes = op[5] if (not op[0].startswith("reduce") or p["dec"]) else op[6]
                acc[k][0] += t1 - t0
                acc[k][1] += 1
                acc[k][2] = nbytes
            spans.append(max(r[2] for r in grp) - min(r[1] for r in grp))
            end, b = -1, 0
            for _, t0, t1 in sorted(grp, key=lambda r: r[1]):   # union of the launches' intervals
                if t1 <= end:
                    continue
                b += t1 - max(t0, end)
                end = t1
            busy.append(b)
        print(f"{'launch':34s} {'octave':>6s} {'R':>3s} {'MB':>8s} {'us':>8s} {'GB/s':>7s} {'frac':>6s}")
        tb = tt = 0.0
        for op, (ns, cnt, nbytes) in zip(ops, acc):
            if not cnt:
                continue
            us = ns / cnt / 1e3
            print(f"{op[0]:34s} {op[1]:6d} {op[3]:3d} {nbytes / 1e6:8.1f} {us:8.1f} {nbytes / us / 1e3:7.0f} {nbytes / us / 1e3 / PEAK:6.3f}")
            tb += nbytes
            tt += us
        print(f"{'family (sum bytes / sum time)':34s} {'':6s} {'':3s} {tb / 1e6:8.1f} {tt:8.1f} {tb / tt / 1e3:7.0f} {tb / tt / 1e3 / PEAK:6.3f}")
        bu = float(np.mean(busy)) / 1e3
        print(f"{'family (sum bytes / busy time)':34s} {'':6s} {'':3s} {tb / 1e6:8.1f} {bu:8.1f} {tb / bu / 1e3:7.0f} {tb / bu / 1e3 / PEAK:6.3f}"
              "   <- bench.py's roofline.frac: time during which at least one blur launch runs (launches of two streams overlap)")
        tail = [(nbytes, ns / cnt / 1e3) for op, (ns, cnt, nbytes) in zip(ops, acc) if cnt and op[1] > 0 or op[0].startswith("reduce")]
        if tail:
            b_, t_ = sum(x[0] for x in tail), sum(x[1] for x in tail)
            print(f"{'launches outside octave 0 (sum time)':34s} {'':5s} {'':3s} {b_ / 1e6:8.1f} {t_:8.1f} {b_ / t_ / 1e3:7.0f} {b_ / t_ / 1e3 / PEAK:6.3f}")
        print(f"pyramid span (first blur start to last blur end), median over {len(spans)} steps: {np.median(spans) / 1e3:.1f} us; "
              f"busy time {np.median(busy) / 1e3:.1f} us; sum of its launches {tt:.1f} us")


if __name__ == "__main__":
    main()
