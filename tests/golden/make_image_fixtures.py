#!/usr/bin/env python3
"""Small image files for the ingest tests (tests/test_cli_io.py), written by an encoder of their own (zlib + struct,
nothing shared with sift_amd/csrc/image_io.cpp): every PNG colour type, bit depths 1-16, all five scanline filters, one
Adam7 file; binary and ASCII PGM / PPM.  Beside them `expected.npz`: what vigra::importImage into a scalar float array
(band 0, unscaled; grey below 8 bit expanded to 0..255) and cv::imread(CV_LOAD_IMAGE_COLOR) (B,G,R 8 bit) give for
each (/root/reference/main.cpp:52-54, :59).

    python tests/golden/make_image_fixtures.py
"""
import os
import struct
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "img")
ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)


def pack_rows(samples, depth):
    """samples [h, w*chans] ints -> list of byte rows at `depth` bits per sample."""
    rows = []
    for r in samples:
        if depth == 8:
            rows.append(bytes(int(v) for v in r))
        elif depth == 16:
            rows.append(b"".join(struct.pack(">H", int(v)) for v in r))
        else:
            bits = "".join(format(int(v), f"0{depth}b") for v in r)
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def filter_rows(rows, bpp, first_filter=0):
    """every row gets filter type (first_filter + y) % 5, so all five occur"""
    out, prev = b"", bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        ft = (first_filter + y) % 5
        f = bytearray()
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][ft]
            f.append((v - pred) & 0xff)
        out += bytes([ft]) + bytes(f)
        prev = row
    return out


def write_png(path, arr, ctype, depth, palette=None, interlace=False):
    """arr [h, w, chans] of sample values"""
    h, w, chans = arr.shape
    bpp = max(1, chans * depth // 8)
    raw = b""
    if interlace:
        for x0, y0, dx, dy in ADAM7:
            sub = arr[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filter_rows(pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, first_filter=x0 + y0)
    else:
        raw = filter_rows(pack_rows(arr.reshape(h, -1), depth), bpp)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(int(v) for v in palette.reshape(-1)))
    z = zlib.compress(raw, 9)
    data += chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")   # two IDAT chunks
    open(path, "wb").write(data)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261002)
    exp = {}

    def expect(name, band0, bgr):
        exp[name + "/band0"] = np.asarray(band0, np.float32)
        exp[name + "/bgr"] = np.asarray(bgr, np.uint8)

    w, h = 23, 17
    rgb = rng.integers(0, 256, (h, w, 3))
    write_png(os.path.join(OUT, "rgb8.png"), rgb, 2, 8)
    expect("rgb8.png", rgb[:, :, 0], rgb[:, :, ::-1])
    rgba = np.concatenate([rgb, rng.integers(0, 256, (h, w, 1))], axis=2)
    write_png(os.path.join(OUT, "rgba8_adam7.png"), rgba, 6, 8, interlace=True)
    expect("rgba8_adam7.png", rgb[:, :, 0], rgb[:, :, ::-1])
    g16 = rng.integers(0, 65536, (h, w, 1))
    write_png(os.path.join(OUT, "grey16.png"), g16, 0, 16)
    expect("grey16.png", g16[:, :, 0], np.repeat(g16 >> 8, 3, axis=2))
    rgb16 = rng.integers(0, 65536, (h, w, 3))
    write_png(os.path.join(OUT, "rgb16.png"), rgb16, 2, 16)
    expect("rgb16.png", rgb16[:, :, 0], (rgb16 >> 8)[:, :, ::-1])
    ga = rng.integers(0, 256, (h, w, 2))
    write_png(os.path.join(OUT, "greyalpha8.png"), ga, 4, 8)
    expect("greyalpha8.png", ga[:, :, 0], np.repeat(ga[:, :, :1], 3, axis=2))
    for depth in (1, 2, 4):
        g = rng.integers(0, 1 << depth, (h, w, 1))
        write_png(os.path.join(OUT, f"grey{depth}.png"), g, 0, depth, interlace=(depth == 2))
        scaled = g * (255 // ((1 << depth) - 1))
        expect(f"grey{depth}.png", scaled[:, :, 0], np.repeat(scaled, 3, axis=2))
    pal = rng.integers(0, 256, (13, 3))
    idx = rng.integers(0, 13, (h, w, 1))
    write_png(os.path.join(OUT, "palette4.png"), idx, 3, 4, palette=pal)
    prgb = pal[idx[:, :, 0]]
    expect("palette4.png", prgb[:, :, 0], prgb[:, :, ::-1])
    # PNM
    g8 = rng.integers(0, 256, (h, w))
    open(os.path.join(OUT, "grey.pgm"), "wb").write(b"P5\n# a comment\n%d %d\n255\n" % (w, h) + bytes(int(v) for v in g8.reshape(-1)))
    expect("grey.pgm", g8, np.repeat(g8[:, :, None], 3, axis=2))
    open(os.path.join(OUT, "grey_ascii.pgm"), "w").write("P2\n%d %d\n255\n" % (w, h) + "\n".join(" ".join(str(int(v)) for v in r) for r in g8) + "\n")
    expect("grey_ascii.pgm", g8, np.repeat(g8[:, :, None], 3, axis=2))
    open(os.path.join(OUT, "rgb.ppm"), "wb").write(b"P6 %d %d 255\n" % (w, h) + bytes(int(v) for v in rgb.reshape(-1)))
    expect("rgb.ppm", rgb[:, :, 0], rgb[:, :, ::-1])
    g12 = rng.integers(0, 4096, (h, w))
    open(os.path.join(OUT, "grey12.pgm"), "wb").write(b"P5 %d %d 4095\n" % (w, h) + b"".join(struct.pack(">H", int(v)) for v in g12.reshape(-1)))
    expect("grey12.pgm", g12, np.repeat((g12 >> 8)[:, :, None], 3, axis=2))
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **exp)
    print("wrote", len(exp) // 2, "files to", OUT)


if __name__ == "__main__":
    main()
