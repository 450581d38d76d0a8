"""Command line mirror of the reference's main.cpp (options main.cpp:28-39, result file :78-89,
overlay :59-76) on top of the GPU path — SURVEY.md §8(f) rows 1-3.

    python -m sift_amd.cli -i image.pgm [-s 1.6] [-k 1.41421354] [-o 4] [-d 3] [-p 0] [-r 1]

Image ingest follows Vigra's scalar import: band 0 of multi-band files, values unscaled.  PGM,
PPM, PNG and JPEG are decoded by the library itself (sift_amd/csrc/image_io.cpp, jpeg_decode.cpp: no PIL, no OpenCV, no libjpeg).
"""
from __future__ import annotations

import argparse
import ctypes as C
import math
import sys

import numpy as np

from . import _lib
from .sift import K_SQRT2, PreconditionViolation, Sift


def _err_call(fn, *args):
    err = C.create_string_buffer(512)
    rc = fn(*args, err, 512)
    if rc:
        raise OSError(err.value.decode(errors="replace") or f"sift_hip image call failed ({rc})")


def image_info(path: str):
    """(width, height, bands, bits per sample) of a PGM / PPM / PNG / JPEG file."""
    L = _lib.load()
    w, h, b, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    _err_call(L.sift_hip_image_info, path.encode(), C.byref(w), C.byref(h), C.byref(b), C.byref(d))
    return w.value, h.value, b.value, d.value


def read_image(path: str) -> np.ndarray:
    """vigra::importImage into a scalar float array (main.cpp:52-54): band 0, values unscaled.  Decoded by the
    library's own PGM / PPM / PNG / JPEG reader (sift_amd/csrc/image_io.cpp), no PIL."""
    L = _lib.load()
    w, h, _, _ = image_info(path)
    out = np.empty((h, w), np.float32)
    _err_call(L.sift_hip_image_read_band0, path.encode(), out.reshape(-1), out.size)
    return out


def read_image_bgr(path: str) -> np.ndarray:
    """cv::imread(path, CV_LOAD_IMAGE_COLOR) (main.cpp:59): [h, w, 3] uint8 in B, G, R order."""
    L = _lib.load()
    w, h, _, _ = image_info(path)
    out = np.empty((h, w, 3), np.uint8)
    _err_call(L.sift_hip_image_read_bgr8, path.encode(), out.reshape(-1), out.size)
    return out


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


def keypoint_records(points) -> np.ndarray:
    kp = np.zeros(len(points), _lib.KEYPOINT_DTYPE)
    for i, p in enumerate(points):
        kp[i] = (p.scale, p.orientation, p.loc[0], p.loc[1], p.octave, p.index, p.filtered, bool(p.descriptors), 0)
    return kp


def overlay_box(kp_record, subpixel: bool):
    """The box main.cpp:60-67 draws for one keypoint: (cx, cy, side, corner points [4, 2])."""
    L = _lib.load()
    rec = np.asarray(kp_record, _lib.KEYPOINT_DTYPE).reshape(1)
    cx, cy, side = C.c_uint16(), C.c_uint16(), C.c_int()
    pts = np.zeros(8, np.float32)
    L.sift_hip_overlay_box(rec.ctypes.data, int(subpixel), C.byref(cx), C.byref(cy), C.byref(side), pts)
    return cx.value, cy.value, side.value, pts.reshape(4, 2)


def draw_overlay(src_path: str, points, subpixel: bool, dst_path: str) -> None:
    """main.cpp:59-76: rotated boxes of side (int)(10*scale) at (loc * 2^octave) / subpixel_divisor on the colour image,
    1-px lines of Scalar(255, 0, 0) (blue in OpenCV's BGR), written as PNG."""
    L = _lib.load()
    img = read_image_bgr(src_path)
    kp = points if isinstance(points, np.ndarray) else keypoint_records(points)
    h, w, _ = img.shape
    if L.sift_hip_overlay_draw(img.reshape(-1), w, h, kp.ctypes.data, kp.size, int(subpixel)):
        raise ValueError("sift_hip_overlay_draw failed")
    _err_call(L.sift_hip_png_write_bgr8, dst_path.encode(), img.reshape(-1), w, h)


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
        if image_info(path)[3] == 8:
            img = img.astype(np.uint8)     # an 8-bit file: its samples cross the link as bytes, the GPU widens them to the same floats
        sift = Sift(args.dogsPerEpoch, args.octaves, args.sigma, args.k, bool(args.subpixel))
        points = sift.calculate(img)
        if not args.no_overlay:
            draw_overlay(path, points, sift.subpixel, path + "_orientation.png")
        if args.result:
            write_result("interstpoints.txt", points)
        print(f"{len(points)} interest points")
    except (PreconditionViolation, AssertionError, RuntimeError, OSError) as ex:  # main.cpp:90-92
        print(ex, file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
