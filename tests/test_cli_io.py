"""SURVEY.md §8(f) rows 1-3 on the CPU: the result file of /root/reference/main.cpp:78-89, the image ingest of
main.cpp:52-54 / :59 and the overlay geometry of main.cpp:60-73 (host code of libsift_hip.so: sift_amd/csrc/image_io.cpp,
sift_amd/cli.py).  No GPU call is made here; the end-to-end runs are in tests/test_gpu_parity.py."""
import hashlib
import math
import os
import struct

import numpy as np
import pytest

import oracle_lib as O
from golden_util import GOLDEN, read_pgm
from sift_amd import _lib, cli
from sift_amd.sift import InterestPoint

IMG = os.path.join(GOLDEN, "img")
# sha256 of interstpoints.txt for the parrot fixture at BASELINE configs[0]'s parameters (3 DoGs, 4 octaves), written by
# the oracle through C++ iostreams exactly as main.cpp:78-89 does (the oracle's points are pinned to the reference binary's)
PARROT_RESULT_SHA = "b12773a011f74da32052d2534991192196e430c841723cd8e9605ccf63918d04"


def oracle_points(run):
    pts, desc = run.points("final")
    return [InterestPoint(float(k["scale"]), int(k["octave"]), int(k["index"]), bool(k["filtered"]), (int(k["x"]), int(k["y"])),
                          float(k["orientation"]), desc[i][:k["n_desc"]].tolist()) for i, k in enumerate(pts)]


def test_result_file_is_byte_exact(tmp_path):
    """cli.write_result (Python "%g") against the C++ `operator<<(float)` text of main.cpp:78-89, whole file."""
    run = O.OracleRun(read_pgm(os.path.join(GOLDEN, "parrot_r.pgm")), 3, 4)
    run.write_result(tmp_path / "expected.txt")
    want = open(tmp_path / "expected.txt", "rb").read()
    assert hashlib.sha256(want).hexdigest() == PARROT_RESULT_SHA
    cli.write_result(str(tmp_path / "got.txt"), oracle_points(run))
    assert open(tmp_path / "got.txt", "rb").read() == want


def test_result_file_special_values(tmp_path):
    """nan / inf / tiny / huge / negative zero print like the C++ stream prints them (glibc "%g")."""
    vals = [float("nan"), -float("nan"), float("inf"), -float("inf"), 0.0, -0.0, 1e-45, 3.4028235e38, 177.49134826660156, 1e-5,
            123456.7, 1234567.0, 0.1]
    p = InterestPoint(1.5, 0, 1, False, (7, 9), 2.5, [float(np.float32(v)) for v in vals])
    cli.write_result(str(tmp_path / "r.txt"), [p])
    line = open(tmp_path / "r.txt").read().splitlines()[1]
    assert line == "[7, 9]\t1.5\t2.5\t[nan, -nan, inf, -inf, 0, -0, 1.4013e-45, 3.40282e+38, 177.491, 1e-05, 123457, 1.23457e+06, 0.1, ]"


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in np.load(os.path.join(IMG, "expected.npz")).files}))
def test_image_reader(name):
    """PGM / PPM / PNG without PIL: band 0 like vigra::importImage into a scalar array (an RGB or palette file gives its RED
    band, App. B-15), and the B,G,R bytes of cv::imread(CV_LOAD_IMAGE_COLOR)."""
    exp = np.load(os.path.join(IMG, "expected.npz"))
    path = os.path.join(IMG, name)
    band0 = cli.read_image(path)
    assert band0.dtype == np.float32 and np.array_equal(band0, exp[name + "/band0"])
    assert np.array_equal(cli.read_image_bgr(path), exp[name + "/bgr"])
    w, h, bands, bits = cli.image_info(path)
    assert (h, w) == band0.shape and bands in (1, 2, 3, 4) and bits in (8, 16)


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in np.load(os.path.join(IMG, "expected_jpeg.npz")).files}))
def test_jpeg_reader(name):
    """JPEG without libjpeg (sift_amd/csrc/jpeg_decode.cpp): pixel for pixel what libjpeg-turbo's default decode returns
    (tests/golden/make_jpeg_fixtures.py): baseline and progressive files, 4:4:4 / 4:2:2 / 4:2:0, restart intervals, custom
    Huffman tables, greyscale, RGB colour space, sizes that are not whole MCUs, chroma planes too narrow for the triangle filter."""
    exp = np.load(os.path.join(IMG, "expected_jpeg.npz"))
    path = os.path.join(IMG, name)
    band0 = cli.read_image(path)
    assert band0.dtype == np.float32 and np.array_equal(band0, exp[name + "/band0"])
    assert np.array_equal(cli.read_image_bgr(path), exp[name + "/bgr"])
    w, h, bands, bits = cli.image_info(path)
    assert (h, w) == band0.shape and bands == (1 if name.startswith("grey") else 3) and bits == 8


def test_jpeg_reader_against_pil_large(tmp_path):
    """Files of the size the command line program is fed, encoded here: every MCU geometry again at 1080p / odd sizes."""
    Image = pytest.importorskip("PIL.Image")
    from sift_amd.synthetic import synth_frame
    g = [synth_frame(1920, 1080, s).astype(np.uint8) for s in (1, 2, 3)]
    cases = [("a.jpg", np.stack(g, 2), dict(quality=90, subsampling=2)),
             ("b.jpg", np.stack(g, 2)[:1079, :1913], dict(quality=75, subsampling=1, progressive=True)),
             ("c.jpg", np.stack(g, 2)[:517, :1001], dict(quality=95, subsampling=0, optimize=True, restart_marker_rows=2)),
             ("d.jpg", g[0][:1001, :777], dict(quality=85, progressive=True))]
    for name, arr, opts in cases:
        Image.fromarray(arr).save(tmp_path / name, "JPEG", **opts)
        ref = np.asarray(Image.open(tmp_path / name).convert("RGB"))
        assert np.array_equal(cli.read_image_bgr(str(tmp_path / name))[:, :, ::-1], ref), name
        assert np.array_equal(cli.read_image(str(tmp_path / name)), ref[:, :, 0].astype(np.float32)), name


def test_jpeg_reader_resyncs_like_libjpeg(tmp_path):
    """Damaged restart markers (a marker missing, renumbered, a whole segment missing, garbage in front of one): libjpeg only warns
    and resynchronises (jdmarker.c jpeg_resync_to_restart, jdhuff.c insufficient_data), so the reference's ingest still returns an
    image; so does this reader, pixel for pixel what libjpeg-turbo returns (through PIL)."""
    import io
    import re
    Image = pytest.importorskip("PIL.Image")
    from sift_amd.synthetic import synth_frame
    img = synth_frame(320, 240, 3).astype(np.uint8)
    rgb = np.stack([img, np.roll(img, 5, 1), np.roll(img, 9, 0)], -1)
    for k, opts in enumerate((dict(restart_marker_rows=1), dict(restart_marker_blocks=3), dict(restart_marker_rows=1, progressive=True),
                              dict(restart_marker_blocks=5, subsampling=0))):
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, "JPEG", quality=85, **opts)
        b = buf.getvalue()
        pos = [m.start() for m in re.finditer(b"\xff[\xd0-\xd7]", b)]
        assert len(pos) > 8
        renum = lambda d: b[:pos[3] + 1] + bytes([0xd0 + ((b[pos[3] + 1] - 0xd0 + d) & 7)]) + b[pos[3] + 2:]   # noqa: E731
        cases = {"intact": b, "marker removed": b[:pos[3]] + b[pos[3] + 2:], "next number": renum(1), "previous number": renum(-1),
                 "far number": renum(4), "segment dropped": b[:pos[3]] + b[pos[4]:], "garbage before a marker": b[:pos[5]] + b"\x12\x34\x56" + b[pos[5]:]}
        for name, data in cases.items():
            path = tmp_path / f"r{k}.jpg"
            path.write_bytes(data)
            ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
            assert np.array_equal(cli.read_image_bgr(str(path))[:, :, ::-1], ref), (opts, name)


def test_headers_cannot_make_the_readers_allocate(tmp_path):
    """A header is untrusted: a file of a few hundred bytes that announces 2^30 pixels (or 2^26, the largest accepted, without
    the data for them) is refused before any buffer of that size exists (no std::bad_alloc, no OOM kill)."""
    import struct
    import zlib

    def png(w, h, idat):
        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", idat) + chunk(b"IEND", b"")
    whole = open(os.path.join(IMG, "rgb_420.jpg"), "rb").read()
    sof = whole.index(b"\xff\xc0")
    files = {"huge.png": png(32768, 32768, zlib.compress(bytes(64))), "big.png": png(8192, 8192, zlib.compress(bytes(64))),
             "huge.pgm": b"P5 1000000 1000000 255\n" + bytes(64), "big.pgm": b"P5 8192 8192 255\n" + bytes(64),
             "big.ppm": b"P3 8192 8192 255\n1 2 3\n", "overflow.pgm": b"P5 99999999999999999999999999 1 255\n" + bytes(8),
             "huge.jpg": whole[:sof + 5] + b"\xff\xff\xff\xff" + whole[sof + 9:], "big.jpg": whole[:sof + 5] + b"\x20\x00\x20\x00" + whole[sof + 9:1200]}
    for name, data in files.items():
        (tmp_path / name).write_bytes(data)
    for name in files:
        with pytest.raises(OSError):
            cli.read_image(str(tmp_path / name))
        with pytest.raises(OSError):
            cli.read_image_bgr(str(tmp_path / name))


def test_jpeg_reader_on_the_reference_example():
    """The reference's own example input (example/parrot.jpg, where the reference tree is present): its red band is the PGM the
    config 1 tests run on (tests/golden/parrot_r.pgm, written with PIL's libjpeg-turbo in round 1)."""
    path = "/root/reference/example/parrot.jpg"
    if not os.path.exists(path):
        pytest.skip("reference tree absent")
    assert np.array_equal(cli.read_image(path), read_pgm(os.path.join(GOLDEN, "parrot_r.pgm")))
    assert cli.image_info(path) == (488, 600, 3, 8)


def test_rgb_file_takes_band_zero():
    exp = np.load(os.path.join(IMG, "expected.npz"))
    bgr = exp["rgb8.png/bgr"]
    assert np.array_equal(cli.read_image(os.path.join(IMG, "rgb8.png")), bgr[:, :, 2].astype(np.float32))   # red, not a luminance mix
    assert not np.array_equal(bgr[:, :, 2], bgr[:, :, 1])


def test_image_reader_against_pil():
    Image = pytest.importorskip("PIL.Image")
    for name in ("rgb8.png", "rgba8_adam7.png", "palette4.png", "grey4.png", "rgb.ppm", "grey.pgm"):
        ref = np.asarray(Image.open(os.path.join(IMG, name)).convert("RGB"))
        assert np.array_equal(cli.read_image_bgr(os.path.join(IMG, name))[:, :, ::-1], ref), name
    assert np.array_equal(cli.read_image(os.path.join(GOLDEN, "parrot_r.pgm")), read_pgm(os.path.join(GOLDEN, "parrot_r.pgm")))


def test_image_reader_errors(tmp_path):
    with pytest.raises(OSError, match="Unable to open"):
        cli.read_image(str(tmp_path / "missing.png"))
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff\xe0" + bytes(32))
    with pytest.raises(OSError, match="JPEG"):
        cli.read_image(str(tmp_path / "x.jpg"))
    whole = open(os.path.join(IMG, "rgb_420.jpg"), "rb").read()
    (tmp_path / "cut.jpg").write_bytes(whole[:300])          # header only: no scan
    with pytest.raises(OSError, match="JPEG"):
        cli.read_image(str(tmp_path / "cut.jpg"))
    sof = whole.index(b"\xff\xc0")
    (tmp_path / "p12.jpg").write_bytes(whole[:sof + 4] + b"\x0c" + whole[sof + 5:])   # 12-bit samples
    with pytest.raises(OSError, match="8-bit"):
        cli.read_image(str(tmp_path / "p12.jpg"))
    (tmp_path / "arith.jpg").write_bytes(whole[:sof + 1] + b"\xc9" + whole[sof + 2:])   # SOF9: arithmetic coding
    with pytest.raises(OSError, match="not decoded"):
        cli.read_image(str(tmp_path / "arith.jpg"))
    (tmp_path / "bad.png").write_bytes(open(os.path.join(IMG, "rgb8.png"), "rb").read()[:200])
    with pytest.raises(OSError):
        cli.read_image(str(tmp_path / "bad.png"))


def test_image_readers_survive_damaged_files(tmp_path):
    """Flipped bytes, truncations and insertions in every fixture file: a reader either decodes something or reports an error
    (OSError with the library's text); it never crashes, hangs or reads out of bounds (the decoders are hand-written)."""
    names = sorted(n for n in os.listdir(IMG) if n.endswith((".jpg", ".png", ".pgm", ".ppm")))
    rng = np.random.default_rng(7)
    decoded = refused = 0
    for it in range(900):
        name = names[it % len(names)]
        b = bytearray(open(os.path.join(IMG, name), "rb").read())
        if it % 3 == 0:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif it % 3 == 1:
            b = b[:int(rng.integers(1, len(b)))]
        else:
            i = int(rng.integers(0, len(b)))
            b[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9))).tolist())
        p = tmp_path / ("f" + os.path.splitext(name)[1])
        p.write_bytes(b)
        try:
            a = cli.read_image(str(p))
            c = cli.read_image_bgr(str(p))
            assert a.ndim == 2 and c.shape == a.shape + (3,)
            decoded += 1
        except OSError:
            refused += 1
    assert decoded > 100 and refused > 100


def rotated_rect_points_restated(cx, cy, w, h, angle):
    """cv::RotatedRect::points of OpenCV 3.2 (modules/core/src/matrix.cpp), float32 arithmetic spelled out in numpy."""
    f = np.float32
    a_ = float(f(angle)) * math.pi / 180.0
    b = f(f(math.cos(a_)) * f(0.5))
    a = f(f(math.sin(a_)) * f(0.5))
    cx, cy, w, h = f(cx), f(cy), f(w), f(h)
    p0 = (f(f(cx - f(a * h)) - f(b * w)), f(f(cy + f(b * h)) - f(a * w)))
    p1 = (f(f(cx + f(a * h)) - f(b * w)), f(f(cy - f(b * h)) - f(a * w)))
    p2 = (f(f(f(2) * cx) - p0[0]), f(f(f(2) * cy) - p0[1]))
    p3 = (f(f(f(2) * cx) - p1[0]), f(f(f(2) * cy) - p1[1]))
    return np.array([p0, p1, p2, p3], np.float32)


def test_rotated_rect_points():
    L = _lib.load()
    rng = np.random.default_rng(5)
    cases = [(100, 50, 18, 18, 0.0), (100, 50, 18, 18, 90.0), (100, 50, 6, 6, 177.49134826660156), (3, 4, 11, 11, -33.3)]
    cases += [tuple(rng.uniform(0, 2000, 2)) + tuple(rng.integers(0, 60, 1).repeat(2)) + (float(rng.uniform(-400, 400)),) for _ in range(200)]
    for cx, cy, w, h, ang in cases:
        got = np.zeros(8, np.float32)
        L.sift_hip_rotated_rect_points(cx, cy, w, h, ang, got)
        assert got.tobytes() == rotated_rect_points_restated(cx, cy, w, h, ang).tobytes(), (cx, cy, w, h, ang)
    # angle 0: bottom-left, top-left, top-right, bottom-right (y grows downwards), OpenCV's documented order
    got = np.zeros(8, np.float32)
    L.sift_hip_rotated_rect_points(100, 50, 18, 10, 0.0, got)
    assert got.reshape(4, 2).tolist() == [[91, 55], [91, 45], [109, 45], [109, 55]]


def kp_record(x, y, octave, scale, orientation):
    kp = np.zeros(1, _lib.KEYPOINT_DTYPE)
    kp[0] = (scale, orientation, x, y, octave, 1, 0, 1, 0)
    return kp


def test_overlay_box_geometry():
    """main.cpp:60-67: u16_t centre (wraps), subpixel divisor, cv::Size's int truncation of scale * 10."""
    cx, cy, side, pts = cli.overlay_box(kp_record(100, 60, 2, 1.8745, 0.0), False)
    assert (cx, cy, side) == (400, 240, 18)                      # 18.745 -> 18
    assert np.array_equal(pts, rotated_rect_points_restated(400, 240, 18, 18, 0.0))
    cx, cy, side, _ = cli.overlay_box(kp_record(101, 61, 1, 0.6627417, 177.49134826660156), True)
    assert (cx, cy, side) == (101, 61, 6)                        # (101 * 2) / 2, 6.627 -> 6
    cx, cy, side, _ = cli.overlay_box(kp_record(101, 61, 0, 0.0, 1.0), True)
    assert (cx, cy, side) == (50, 30, 0)                         # 50.5 -> 50: the division is in double, the store truncates
    cx, cy, _, _ = cli.overlay_box(kp_record(40000, 9000, 3, 1.0, 0.0), False)
    assert (cx, cy) == ((40000 * 8) % 65536, (9000 * 8) % 65536)  # u16_t x, y: 320000 -> 57856, 72000 -> 6464


def bresenham_reference(x1, y1, x2, y2):
    """8-connected line, left to right, error term of OpenCV's LineIterator (drawing.cpp), written independently as a
    pixel generator."""
    if x2 < x1:
        x1, y1, x2, y2 = x2, y2, x1, y1
    dx, dy = x2 - x1, abs(y2 - y1)
    sy = 1 if y2 >= y1 else -1
    pts, x, y = [], x1, y1
    if dy <= dx:
        err = dx - 2 * dy
        for _ in range(dx + 1):
            pts.append((x, y))
            if err < 0:
                y += sy
                err += 2 * dx
            err -= 2 * dy
            x += 1
    else:
        err = dy - 2 * dx
        for _ in range(dy + 1):
            pts.append((x, y))
            if err < 0:
                x += 1
                err += 2 * dy
            err -= 2 * dx
            y += sy
    return pts


def test_overlay_matches_the_independent_renderer_on_random_boxes():
    """sift_hip_overlay_draw against tests/overlay_ref.py on 150 images of 20 random boxes each, most of which cross the image
    border (clipped lines walked from the clipped end points), with and without the subpixel divisor: pixel for pixel."""
    import overlay_ref
    L = _lib.load()
    rng = np.random.default_rng(3)
    w, h = 200, 150
    clipped = 0
    for it in range(150):
        n = 20
        kps = np.zeros(n, _lib.KEYPOINT_DTYPE)
        kps["x"] = rng.integers(0, 230, n)
        kps["y"] = rng.integers(0, 180, n)
        kps["octave"] = rng.integers(0, 2, n)
        kps["scale"] = rng.uniform(0, 9, n).astype(np.float32)
        kps["orientation"] = rng.uniform(-400, 400, n).astype(np.float32)
        if it % 10 == 0:
            kps["orientation"][:3] = (np.nan, np.inf, 177.49134826660156)
        sub = it % 2
        got = np.full((h, w, 3), 7, np.uint8)
        assert L.sift_hip_overlay_draw(got.reshape(-1), w, h, kps.ctypes.data, n, sub) == 0
        want = overlay_ref.draw_overlay(np.full((h, w, 3), 7, np.uint8), kps, bool(sub))
        assert np.array_equal(got, want), it
        clipped += int((want[0] != 7).any() or (want[-1] != 7).any() or (want[:, 0] != 7).any() or (want[:, -1] != 7).any())
    assert clipped > 100


def test_overlay_draw_and_png_roundtrip(tmp_path):
    L = _lib.load()
    w, h = 160, 120
    img = np.full((h, w, 3), 7, np.uint8)
    kps = np.concatenate([kp_record(80, 60, 0, 3.2, 30.0), kp_record(5, 5, 0, 4.0, 177.49134826660156),   # the second box leaves the image
                          kp_record(40, 30, 1, 1.2, float("nan"))])                                          # NaN orientation: nothing drawn
    assert L.sift_hip_overlay_draw(img.reshape(-1), w, h, kps.ctypes.data, kps.size, 0) == 0
    # the whole image, boxes that leave it included, against the independent renderer (tests/overlay_ref.py: cv::clipLine, then
    # LineIterator from the CLIPPED end points) - exact
    import overlay_ref
    want = overlay_ref.draw_overlay(np.full((h, w, 3), 7, np.uint8), kps, False)
    assert np.array_equal(img, want)
    # ... which inside the image is the plain 8-connected walk between the rounded corners
    plain = np.full((h, w, 3), 7, np.uint8)
    _, _, _, p = cli.overlay_box(kps[0], False)
    q = [(int(np.rint(x)), int(np.rint(y))) for x, y in p]        # cvRound: nearest, ties to even
    for a, b in ((0, 1), (0, 3), (2, 3), (1, 2)):
        for x, y in bresenham_reference(*q[a], *q[b]):
            plain[y, x] = (255, 0, 0)
    inside = np.zeros((h, w), bool)
    inside[20:100, 40:120] = True        # the first box lies fully inside
    assert np.array_equal(img[inside], plain[inside])
    drawn = (img != 7).any(axis=2)
    assert drawn[inside].sum() > 100 and drawn[:28, :28].sum() > 40    # the clipped box still leaves its inner parts in the corner
    assert set(map(tuple, img[drawn])) == {(255, 0, 0)}             # B, G, R = Scalar(255, 0, 0)
    # imwrite -> imread round trip through the PNG writer and reader
    path = str(tmp_path / "o.png")
    cli._err_call(L.sift_hip_png_write_bgr8, path.encode(), img.reshape(-1), w, h)
    assert np.array_equal(cli.read_image_bgr(path), img)
    assert open(path, "rb").read(8) == b"\x89PNG\r\n\x1a\n" and struct.unpack(">II", open(path, "rb").read(24)[16:]) == (w, h)
