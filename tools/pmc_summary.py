"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv output per kernel (bytes per launch)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
res = defaultdict(lambda: {"launches": 0, "FETCH_SIZE_KB": 0.0, "WRITE_SIZE_KB": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, f"pmc_{c}", "**", "*counter_collection.csv"), recursive=True)
    seen = defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != c:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sift_hip::", "")
            res[name][c + "_KB"] += float(r["Counter_Value"])
            seen[name].add(r.get("Dispatch_Id"))
    for name, s in seen.items():
        res[name]["launches"] = max(res[name]["launches"], len(s))
out = {}
for name, v in sorted(res.items(), key=lambda kv: -(kv[1]["FETCH_SIZE_KB"] + kv[1]["WRITE_SIZE_KB"])):
    n = max(v["launches"], 1)
    # gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams -> x2 (guide §HBM)
    read = v["FETCH_SIZE_KB"] * 1024 * 2
    write = v["WRITE_SIZE_KB"] * 1024
    out[name] = {"launches": n, "read_bytes_per_launch_x2_corrected": read / n, "write_bytes_per_launch": write / n,
                 "hbm_bytes_per_launch": (read + write) / n}
blur = {k: v for k, v in out.items() if k.startswith(("blur_fused_kernel", "blur_stream_kernel", "blur_stream2_kernel", "blur_reduce_kernel", "blur_pair_kernel"))}
if blur:
    tot_l = sum(v["launches"] for v in blur.values())
    out["_blur_fused_all"] = {"launches": tot_l,
                              "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["launches"] for v in blur.values()) / tot_l}
print(json.dumps(out, indent=1))
