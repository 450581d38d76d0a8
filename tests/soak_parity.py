"""Randomised parity soak: the whole pipeline on shapes, parameters and batch sizes drawn at random, every Gaussian / DoG
level, stage list, orientation and descriptor against the CPU oracle bit for bit (tests/test_gpu_parity.py: compare_run).

Not collected by `pytest tests/` (the file name does not match): a run takes as long as it is given.  The fixed cases of
the suite were chosen for the kernels' known boundaries; this draws the rest of the space - widths of every residue
modulo 2 / 4 / 64 / 128, strips that end inside the image, chunks of odd row counts, levels close to the smallest the
reference accepts, batches of 1 - 3.
    python3 tests/soak_parity.py [seconds=600] [seed=1] [big]
("big": shapes 1000..4096 x 700..2200, batches of 1 - 6 - launches large enough to take the streaming kernels by themselves.)
Prints one line per case and a summary; exits 1 if any case differed.
"""
import os
import sys
import tempfile
import time
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p_ in (os.path.join(HERE, "golden"), HERE, os.path.dirname(HERE)):   # what tests/conftest.py puts on the path
    sys.path.insert(0, p_)

from sift_amd import _lib                                  # noqa: E402
from sift_amd.sift import Context                          # noqa: E402
from sift_amd.synthetic import blob_frame, synth_frame     # noqa: E402
from test_gpu_parity import compare_run                    # noqa: E402


def draw_big(rng):
    w = int(rng.integers(1000, 4097))
    h = int(rng.integers(700, 2201))
    if rng.random() < 0.5:
        w = w // 4 * 4
    frames = int(rng.choice([1, 2, 3, 4, 6])) if w * h < 3000 * 1000 else int(rng.choice([1, 2]))
    return dict(w=w, h=h, dogs=int(rng.choice([3, 3, 4])), octaves=int(rng.integers(2, 6)), sigma=float(rng.choice([1.6, 1.6, 1.2])), subpixel=False,
                frames=frames, streaming=0, blobs=False, seed=int(rng.integers(1, 1 << 20)))


def draw(rng):
    subpixel = rng.random() < 0.1
    big = rng.random() < 0.25
    w = int(rng.integers(48, 700 if subpixel else (1500 if big else 700)))
    h = int(rng.integers(48, 500 if subpixel else (1000 if big else 500)))
    if rng.random() < 0.3:
        w = w // 2 * 2          # the even widths the row-packed streaming blur takes
    dogs = int(rng.choice([3, 3, 3, 4, 5]))
    sigma = float(rng.choice([1.6, 1.6, 1.2, 2.0]))
    side = min(w, h) * (2 if subpixel else 1)
    max_oct = 1
    while max_oct < 5 and (side >> max_oct) >= 40:
        max_oct += 1
    octaves = int(rng.integers(1, max_oct + 1))
    frames = int(rng.choice([1, 1, 2, 3]))
    streaming = int(rng.random() < 0.5)
    blobs = rng.random() < 0.1 and w * h <= 200 * 1000
    seed = int(rng.integers(1, 1 << 20))
    return dict(w=w, h=h, dogs=dogs, octaves=octaves, sigma=sigma, subpixel=subpixel, frames=frames, streaming=streaming, blobs=blobs, seed=seed)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    ctx = Context(0)
    report_dir = tempfile.mkdtemp(prefix="soak_")
    t_end = time.time() + budget
    done, failed, skipped = 0, [], 0
    while time.time() < t_end:
        c = draw_big(rng) if big else draw(rng)
        name = "soak %(w)dx%(h)d dogs %(dogs)d oct %(octaves)d sigma %(sigma)g sub %(subpixel)d x%(frames)d stream %(streaming)d blobs %(blobs)d seed %(seed)d" % c
        img = blob_frame(c["w"], c["h"], 3 + c["seed"] % 5) if c["blobs"] else synth_frame(c["w"], c["h"], c["seed"])
        ctx.set_option("stream_min_waves", 1 if c["streaming"] else 0)
        t0 = time.time()
        try:
            rep = compare_run(ctx, img, c["dogs"], c["octaves"], c["subpixel"], name, report_dir, batch_of=c["frames"], sigma=c["sigma"])
            print("ok   %-110s final %6d  %.1f s" % (name, rep["final"], time.time() - t0), flush=True)
            done += 1
        except AssertionError as e:
            msg = str(e).split("\n")[0][:300]
            if msg and not msg.startswith("soak "):   # compare_run's own messages start with the case's name; anything else is the oracle's error text: the reference refuses this input
                skipped += 1
                print("skip %-110s %s" % (name, msg), flush=True)
            else:
                failed.append((name, msg))
                print("FAIL %-110s %s" % (name, msg), flush=True)
        except Exception:
            failed.append((name, traceback.format_exc().splitlines()[-1]))
            print("FAIL %-110s %s" % (name, failed[-1][1]), flush=True)
            ctx = Context(0)
    print("soak: %d cases bit-identical, %d skipped (the reference refuses the input), %d FAILED" % (done, skipped, len(failed)))
    for name, msg in failed:
        print("  ", name, "::", msg)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
