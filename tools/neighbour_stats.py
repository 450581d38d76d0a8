"""How many preceding neighbours a keypoint's descriptor window has (sift.cpp:80-92 adds every earlier keypoint's
weights into the shared maps: descriptor_wave_kernel walks them), and how much of the window each one covers.
For the bench's frames: per keypoint n_prev = earlier keypoints of the same level within +-15 px in x and y; per
(keypoint, neighbour) pair the covered share of the 16x16 window and the number of 8x8 quarter-windows touched.
    python3 tools/neighbour_stats.py [frames=4]
"""
import sys
import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from sift_amd import _lib                      # noqa: E402
from sift_amd.sift import Context              # noqa: E402
from sift_amd.synthetic import synth_frame     # noqa: E402

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
frames = np.stack([synth_frame(1920, 1080, s + 1) for s in range(nf)])
ctx = Context(0)
ctx.calculate_batch(frames, _lib.Params(3, 4, 1.6, 2 ** 0.5, 0))
counts = ctx.counts()
kp, _ = ctx.results()
off = 0
all_prev, cover, blocks, blocks_x = [], [], [], []
for f in range(nf):
    k = kp[off:off + int(counts[f])]
    off += int(counts[f])
    k = k[k["has_descriptor"] != 0]
    for o in np.unique(k["octave"]):
        for i in np.unique(k["index"]):
            sel = np.nonzero((k["octave"] == o) & (k["index"] == i))[0]      # vector order within the level
            x = k["x"][sel].astype(np.int32)
            y = k["y"][sel].astype(np.int32)
            n = len(sel)
            if n == 0:
                continue
            order = np.argsort(x, kind="stable")
            xs = x[order]
            for a in range(n):
                lo, hi = np.searchsorted(xs, x[a] - 15), np.searchsorted(xs, x[a] + 15, side="right")
                cand = order[lo:hi]
                cand = cand[(cand < a) & (np.abs(y[cand] - y[a]) <= 15)]
                all_prev.append(len(cand))
                if len(cand):
                    dx, dy = x[a] - x[cand], y[a] - y[cand]
                    cover.append((16 - np.abs(dx)) * (16 - np.abs(dy)) / 256.0)
                    bx = (dx >= -7).astype(int) + (dx <= 7).astype(int)
                    by = (dy >= -7).astype(int) + (dy <= 7).astype(int)
                    blocks.append(bx * by)
                    blocks_x.append(np.minimum(4, (16 - np.abs(dx) + 3) // 4 + 1))
p = np.array(all_prev)
c = np.concatenate(cover)
b = np.concatenate(blocks)
print(f"{nf} frames, {len(p)} keypoints with a descriptor: preceding neighbours per keypoint mean {p.mean():.2f}, median {np.median(p):.0f}, "
      f"90 % {np.percentile(p, 90):.0f}, 99 % {np.percentile(p, 99):.0f}, max {p.max()}; none: {np.mean(p == 0):.3f}")
print("histogram of n_prev (0, 1-2, 3-4, 5-8, 9-16, 17-32, 33-64, 65+):",
      [int(((p >= lo) & (p <= hi)).sum()) for lo, hi in ((0, 0), (1, 2), (3, 4), (5, 8), (9, 16), (17, 32), (33, 64), (65, 10 ** 9))])
print(f"{len(c)} (keypoint, neighbour) pairs: covered share of the window mean {c.mean():.3f}; 8x8 quarters touched mean {b.mean():.2f} of 4 "
      f"(1: {np.mean(b == 1):.3f}, 2: {np.mean(b == 2):.3f}, 4: {np.mean(b == 4):.3f})")
print(f"share of pairs with dx == 0 and dy == 0 (same pixel, another orientation): {np.mean(c == 1.0):.3f}")
