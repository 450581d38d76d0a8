#!/usr/bin/env python3
"""Kernels of a rocprofv3 kernel trace inside a time window near the end of the run, with their queues:
   timeline_window.py trace.csv [window_us] [skip_tail_us]   (phase-level view: consecutive blur kernels are merged)"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
win = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 12e6
skip = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 4e6
short = lambda n: re.sub(r"[<(].*", "", n.replace("sift_hip::", "").replace("void ", ""))[:28]
t_end = int(rows[-1]["End_Timestamp"]) - skip
t_beg = t_end - win
sel = [r for r in rows if t_beg <= int(r["Start_Timestamp"]) <= t_end]
out = []
for r in sel:
    n, q = short(r["Kernel_Name"]), r["Queue_Id"]
    s, e = (int(r["Start_Timestamp"]) - t_beg) / 1e3, (int(r["End_Timestamp"]) - t_beg) / 1e3
    if out and out[-1][0] == n and out[-1][1] == q and n.startswith(("blur_", "extrema_edge", "__amd")):
        out[-1][3] = e
        out[-1][4] += 1
    else:
        out.append([n, q, s, e, 1])
for n, q, s, e, k in out:
    print(f"{n:30s} q{q:>3s} {s:9.1f} {e:9.1f} {e - s:8.1f}  x{k}")
