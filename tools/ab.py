#!/usr/bin/env python3
"""A/B on ONE box under a fixed protocol (VERDICT r05 task 6a): two arms of the bench's headline loop run ALTERNATELY,
at least five alternations, every arm summarised by the median and the min - max of its runs' ms_per_step (each run's
headline AND its four in-run repeats count as samples); a winner is only named when the arms' [min, max] intervals do
not overlap - otherwise the line says so and no decision may rest on it.

  python3 tools/ab.py --lib libsift_hip_base.so                 # the other build of the library (a file in sift_amd/lib/)
  python3 tools/ab.py --option pair_waves 1536 2048             # one library option at two values
  python3 tools/ab.py --option reduce_kept 1 0 --diag           # an option only libsift_hip_diag.so knows
  ... [--rounds 5] [--depths 2,1] [--steps 20] [--bench-args "--rccl-loopback"]

Boxes of the pool differ by 3 - 4 % in the headline, so arms are never compared across gpurun calls.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_line(extra, env_extra, steps):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "5", "--no-extras", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        return None, r.stderr[-400:]
    return json.loads(lines[-1]), ""


def summarise(samples):
    return statistics.median(samples), min(samples), max(samples)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", help="file name of the other build in sift_amd/lib/ (arm B); arm A is libsift_hip.so")
    ap.add_argument("--option", nargs=3, metavar=("NAME", "A", "B"))
    ap.add_argument("--diag", action="store_true", help="both arms on libsift_hip_diag.so (options only that build knows)")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--depths", default="2,1")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--bench-args", default="")
    a = ap.parse_args()
    if a.rounds < 5:
        print("ab.py: fewer than 5 alternations resolve nothing on this pool (in-run repeats spread by 2 %): --rounds raised to 5")
        a.rounds = 5
    base_env = {"SIFT_HIP_LIBRARY": "libsift_hip_diag.so"} if a.diag else {}
    if a.lib:
        arms = [("this build", [], dict(base_env)), (a.lib, [], {"SIFT_HIP_LIBRARY": os.path.basename(a.lib)})]
    elif a.option:
        n, va, vb = a.option
        arms = [(f"{n}={va}", ["--set", f"{n}={va}"], dict(base_env)), (f"{n}={vb}", ["--set", f"{n}={vb}"], dict(base_env))]
    else:
        ap.error("--lib or --option")
    extra = a.bench_args.split()
    for depth in [int(d) for d in a.depths.split(",")]:
        samples = {name: [] for name, _, _ in arms}
        fracs = {name: [] for name, _, _ in arms}
        for r in range(a.rounds):
            for name, opts, env in arms:
                d, err = bench_line(["--pipeline-depth", str(depth)] + opts + extra, env, a.steps)
                if d is None:
                    print(f"depth {depth} round {r + 1} {name}: no bench line ({err.strip()[-200:]})")
                    continue
                samples[name] += d["ms_per_step_repeats"]["all"]
                fracs[name].append(d["roofline"]["frac"])
                print(f"depth {depth} round {r + 1} {name:>24}: {d['ms_per_step']:.3f} ms/step  repeats "
                      + " ".join(f"{v:.3f}" for v in d["ms_per_step_repeats"]["all"]) + f"  blur frac {d['roofline']['frac']:.3f}", flush=True)
        (na, _, _), (nb, _, _) = arms
        if not samples[na] or not samples[nb]:
            print(f"depth {depth}: an arm has no samples - nothing to compare")
            continue
        ma, la, ha = summarise(samples[na])
        mb, lb, hb = summarise(samples[nb])
        print(f"depth {depth} SUMMARY {na}: median {ma:.3f} [{la:.3f} - {ha:.3f}] ms/step over {len(samples[na])} samples, blur frac median {statistics.median(fracs[na]):.3f}")
        print(f"depth {depth} SUMMARY {nb}: median {mb:.3f} [{lb:.3f} - {hb:.3f}] ms/step over {len(samples[nb])} samples, blur frac median {statistics.median(fracs[nb]):.3f}")
        if ha < lb:
            print(f"depth {depth} VERDICT: {na} is faster (intervals disjoint; medians differ by {(mb - ma) / mb * 100:.1f} %)")
        elif hb < la:
            print(f"depth {depth} VERDICT: {nb} is faster (intervals disjoint; medians differ by {(ma - mb) / ma * 100:.1f} %)")
        else:
            print(f"depth {depth} VERDICT: UNRESOLVED - the arms' [min, max] intervals overlap (medians differ by {abs(ma - mb) / max(ma, mb) * 100:.1f} %): no 'kept' may rest on this run")


if __name__ == "__main__":
    main()
