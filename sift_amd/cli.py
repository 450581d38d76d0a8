"""Command line mirror of the reference's main.cpp (options main.cpp:28-39, result file :78-89,
overlay :59-76) on top of the GPU path — SURVEY.md §8(f) rows 1-3.

    python -m sift_amd.cli -i image.pgm [-s 1.6] [-k 1.41421354] [-o 4] [-d 3] [-p 0] [-r 1]

Image ingest follows Vigra's scalar import: band 0 of multi-band files, values
0..255 unscaled.  Needs PIL for anything but binary PGM.
"""
from __future__ import annotations

import argparse
import math
import sys

import numpy as np

from .sift import K_SQRT2, PreconditionViolation, Sift


def read_image(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        head = f.read(2)
    if head == b"P5":
        with open(path, "rb") as f:
            assert f.readline().strip() == b"P5"
            line = f.readline()
            while line.startswith(b"#"):
                line = f.readline()
            w, h = map(int, line.split())
            maxv = int(f.readline())
            data = np.frombuffer(f.read(), np.uint8 if maxv < 256 else ">u2").reshape(h, w)
        return np.ascontiguousarray(data, dtype=np.float32)
    from PIL import Image
    img = np.asarray(Image.open(path))
    if img.ndim == 3:
        img = img[:, :, 0]  # band 0, like vigra::importImage into a scalar array
    return np.ascontiguousarray(img, dtype=np.float32)


def fmt(v: float) -> str:
    """operator<<(ostream&, float) with the default precision of 6 significant digits."""
    if math.isnan(v):
        return "nan" if not math.copysign(1.0, v) < 0 else "-nan"
    if math.isinf(v):
        return "inf" if v > 0 else "-inf"
    s = "%g" % v
    return s


def write_result(path: str, points) -> None:
    with open(path, "w") as out:
        out.write("Location\tscale\torientation\tdescriptors\n")
        for p in points:
            out.write(f"[{p.loc[0]}, {p.loc[1]}]\t{fmt(p.scale)}\t{fmt(p.orientation)}\t[")
            for d in p.descriptors:
                out.write(fmt(d) + ", ")
            out.write("]\n")


def draw_overlay(src_path: str, points, subpixel: bool, dst_path: str) -> None:
    """Rotated boxes of side 10*scale at (loc * 2^octave) / subpixel_divisor, 1-px blue lines
    (main.cpp:59-76; OpenCV's Scalar(255,0,0) is BGR blue)."""
    from PIL import Image, ImageDraw
    img = Image.open(src_path).convert("RGB")
    draw = ImageDraw.Draw(img)
    div = 2 if subpixel else 1
    for p in points:
        x = int((p.loc[0] * 2 ** p.octave) / div) & 0xFFFF
        y = int((p.loc[1] * 2 ** p.octave) / div) & 0xFFFF
        half = int(p.scale * 10) / 2.0
        a = math.radians(p.orientation) if not math.isnan(p.orientation) else 0.0
        ca, sa = math.cos(a), math.sin(a)
        corners = [(-half, -half), (half, -half), (half, half), (-half, half)]
        pts = [(x + cx * ca - cy * sa, y + cx * sa + cy * ca) for cx, cy in corners]
        draw.line(pts + [pts[0]], fill=(0, 0, 255), width=1)
    img.save(dst_path)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="sift", description="SIFT features on the GPU (snowiow/SIFT options)")
    ap.add_argument("img_pos", nargs="?", help="image (positional, like the reference)")
    ap.add_argument("--img", "-i", help="The image on which sift will be executed")
    ap.add_argument("--sigma", "-s", type=float, default=1.6)
    ap.add_argument("--k", "-k", type=float, default=K_SQRT2)
    ap.add_argument("--octaves", "-o", type=int, default=4)
    ap.add_argument("--dogsPerEpoch", "-d", type=int, default=3)
    ap.add_argument("--subpixel", "-p", type=int, default=0)
    ap.add_argument("--result", "-r", type=int, default=0)
    ap.add_argument("--no-overlay", action="store_true", help="skip <img>_orientation.png")
    args = ap.parse_args(argv)
    path = args.img or args.img_pos
    if not path:
        ap.print_help()
        return 1
    try:
        img = read_image(path)
        sift = Sift(args.dogsPerEpoch, args.octaves, args.sigma, args.k, bool(args.subpixel))
        points = sift.calculate(img)
        if not args.no_overlay:
            try:
                draw_overlay(path, points, sift.subpixel, path + "_orientation.png")
            except ImportError:
                print("PIL not available: overlay skipped", file=sys.stderr)
        if args.result:
            write_result("interstpoints.txt", points)
        print(f"{len(points)} interest points")
    except (PreconditionViolation, AssertionError, RuntimeError, OSError) as ex:  # main.cpp:90-92
        print(ex, file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
