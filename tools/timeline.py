#!/usr/bin/env python3
"""Timeline of the last bench step from a rocprofv3 kernel trace: start/end (us) relative to the step's
first kernel, per kernel and queue, plus the idle gaps on the union of all queues."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(.*", "", n.replace("sift_hip::", "").replace("void ", ""))[:44]
# a step starts at the first-level blur of the base image: find the last occurrence of the step's first kernel
names = [short(r["Kernel_Name"]) for r in rows]
first = next(i for i in range(len(rows) - 1, -1, -1) if names[i].startswith("blur_") and (i == 0 or not names[i - 1].startswith(("blur_", "resample"))) and
             all(not n.startswith("descriptor") for n in names[max(0, i - 3):i]) or i == 0)
# simpler: last kernel named descriptor_kernel ends a step; walk back to the previous descriptor_kernel
desc = [i for i, n in enumerate(names) if n.startswith(("descriptor_kernel", "descriptor_wave_kernel"))]
end = desc[-1]
begin = desc[-2] + 1 if len(desc) > 1 else 0
t0 = int(rows[begin]["Start_Timestamp"])
last_end = t0
print(f"{'kernel':46s} {'queue':>5s} {'start':>9s} {'end':>9s} {'dur':>8s}   gap-before(all queues)")
for r, n in zip(rows[begin:end + 1], names[begin:end + 1]):
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = s - (last_end - t0)
    print(f"{n:46s} {r['Queue_Id']:>5s} {s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}   {gap/1e3:7.1f}" if gap > 5000 else
          f"{n:46s} {r['Queue_Id']:>5s} {s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}")
    last_end = max(last_end, int(r["End_Timestamp"]))
print("step span (us):", (last_end - t0) / 1e3)
