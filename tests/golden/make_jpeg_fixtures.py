#!/usr/bin/env python3
"""Small JPEG files for the ingest tests (tests/test_cli_io.py) and what libjpeg-turbo returns for them.

Encoder and reference decoder are PIL's bundled libjpeg-turbo in the build container (default decode settings: islow IDCT,
fancy upsampling — what vigra::importImage and cv::imread get from libjpeg, /root/reference/main.cpp:52-54, :59).  Beside
the files `expected_jpeg.npz`: band 0 as float (red of a colour file, App. B-15) and the B,G,R bytes for each.

    python tests/golden/make_jpeg_fixtures.py
"""
import io
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "img")


def picture(rng, w, h, chans):
    """smooth ramps + blobs + noise: exercises DC prediction, long zero runs and the range limit at once"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.zeros((h, w, chans))
    for c in range(chans):
        a = 128 + 110 * np.sin(x / (3.0 + c) + c) * np.cos(y / (4.0 + 2 * c))
        a += rng.normal(0, 25 + 10 * c, (h, w))
        a[h // 3:h // 2, w // 4:w // 2] = 255 * (c % 2)      # hard edges into saturation
        out[:, :, c] = a
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261003)
    exp = {}
    cases = [
        # name, w, h, mode, save options
        ("rgb_420.jpg", 37, 29, "RGB", dict(quality=85, subsampling=2)),
        ("rgb_422.jpg", 40, 21, "RGB", dict(quality=75, subsampling=1)),
        ("rgb_444.jpg", 19, 33, "RGB", dict(quality=92, subsampling=0)),
        ("rgb_420_q100.jpg", 33, 18, "RGB", dict(quality=100, subsampling=2)),
        ("rgb_420_q10.jpg", 50, 41, "RGB", dict(quality=10, subsampling=2)),
        ("rgb_420_optimized.jpg", 29, 30, "RGB", dict(quality=80, subsampling=2, optimize=True)),
        ("rgb_420_progressive.jpg", 45, 38, "RGB", dict(quality=85, subsampling=2, progressive=True)),
        ("rgb_444_progressive.jpg", 26, 35, "RGB", dict(quality=60, subsampling=0, progressive=True)),
        ("rgb_422_progressive_restart.jpg", 41, 27, "RGB", dict(quality=70, subsampling=1, progressive=True, restart_marker_blocks=2)),
        ("rgb_420_restart.jpg", 64, 48, "RGB", dict(quality=85, subsampling=2, restart_marker_blocks=3)),
        ("rgb_420_restart_rows.jpg", 35, 50, "RGB", dict(quality=85, subsampling=2, restart_marker_rows=1)),
        ("grey.jpg", 31, 27, "L", dict(quality=85)),
        ("grey_progressive.jpg", 24, 40, "L", dict(quality=50, progressive=True)),
        ("rgb_420_1x1.jpg", 1, 1, "RGB", dict(quality=90, subsampling=2)),
        ("rgb_420_3x2.jpg", 3, 2, "RGB", dict(quality=90, subsampling=2)),     # chroma 2 samples wide: plain replication
        ("rgb_420_5x5.jpg", 5, 5, "RGB", dict(quality=90, subsampling=2)),     # chroma 3 wide: triangle filter
        ("rgb_422_4x3.jpg", 4, 3, "RGB", dict(quality=90, subsampling=1)),
        ("rgb_420_16x16.jpg", 16, 16, "RGB", dict(quality=85, subsampling=2)),
        ("rgb_420_17x17.jpg", 17, 17, "RGB", dict(quality=85, subsampling=2)),
        ("rgb_420_wide.jpg", 131, 9, "RGB", dict(quality=85, subsampling=2)),
    ]
    for name, w, h, mode, opts in cases:
        arr = picture(rng, w, h, 3 if mode == "RGB" else 1)
        im = Image.fromarray(arr if mode == "RGB" else arr[:, :, 0], mode)
        path = os.path.join(OUT, name)
        im.save(path, "JPEG", **opts)
        dec = Image.open(path)
        rgb = np.asarray(dec.convert("RGB"))
        raw = np.asarray(dec)
        band0 = raw if raw.ndim == 2 else raw[:, :, 0]
        exp[name + "/band0"] = band0.astype(np.float32)
        exp[name + "/bgr"] = np.ascontiguousarray(rgb[:, :, ::-1])
    # a frame large enough for the whole path (4 octaves): the CLI test feeds it through sift_amd.cli
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from sift_amd.synthetic import synth_frame
    g = [synth_frame(320, 240, s).astype(np.uint8) for s in (41, 42, 43)]
    path = os.path.join(OUT, "scene_420.jpg")
    Image.fromarray(np.stack(g, axis=2), "RGB").save(path, "JPEG", quality=90, subsampling=2)
    rgb = np.asarray(Image.open(path).convert("RGB"))
    exp["scene_420.jpg/band0"] = rgb[:, :, 0].astype(np.float32)
    exp["scene_420.jpg/bgr"] = np.ascontiguousarray(rgb[:, :, ::-1])
    # RGB colour space inside the file (no YCbCr transform), where this PIL can write it
    try:
        arr = picture(rng, 22, 20, 3)
        path = os.path.join(OUT, "rgb_keep_rgb.jpg")
        Image.fromarray(arr, "RGB").save(path, "JPEG", quality=90, keep_rgb=True)
        rgb = np.asarray(Image.open(path).convert("RGB"))
        exp["rgb_keep_rgb.jpg/band0"] = rgb[:, :, 0].astype(np.float32)
        exp["rgb_keep_rgb.jpg/bgr"] = np.ascontiguousarray(rgb[:, :, ::-1])
    except Exception as e:   # pragma: no cover
        print("keep_rgb not written:", e)
    np.savez_compressed(os.path.join(OUT, "expected_jpeg.npz"), **exp)
    print("wrote", len(exp) // 2, "JPEG files to", OUT)


if __name__ == "__main__":
    main()
