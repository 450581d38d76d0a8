#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the development container).

The reference has no tests or golden vectors of its own, and cannot be built
here, so these vectors come from the oracle (oracle/) — they pin the oracle AND the HIP path
against regressions and travel to the GPU box, where /root/reference does not exist.
  parrot_r.pgm     R band of /root/reference/example/parrot.jpg decoded with PIL here (BASELINE
                   config 1 input; Vigra's scalar import takes band 0)
  kats.npz         tap tables, resampling index maps, vertexParabola values
  case_*.npz       input seed/shape, per-level SHA-256 of the pyramid, candidate count and flags
                   digest, final keypoints and descriptors
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from sift_amd.synthetic import synth_frame  # noqa: E402

SIGMAS = [1.0, 1.6, 2.2627418, 3.2, 4.5254836, 6.4, 9.050967, 12.8, 18.101934]
LUTS = [(1920, 960), (1080, 540), (135, 68), (488, 244), (600, 300), (240, 120), (1920, 3840), (7, 4)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def write_pgm(path, a):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
        f.write(a.astype(np.uint8).tobytes())


def read_pgm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P5"
        w, h = map(int, f.readline().split())
        assert int(f.readline()) == 255
        return np.frombuffer(f.read(), np.uint8).reshape(h, w).astype(np.float32)


def make_case(name, img, dogs, octaves, subpixel=False, meta=None):
    run = O.OracleRun(img, dogs, octaves, subpixel=subpixel)
    assert run.status == 0, run.error
    levels = {}
    for o in range(octaves):
        for j in range(dogs + 1):
            levels[f"g{o}_{j}"] = sha(run.level("gaussian", o, j))
        for j in range(dogs):
            levels[f"d{o}_{j}"] = sha(run.level("dog", o, j))
    cand, _ = run.points("candidates")
    fin, desc = run.points("final")
    np.savez_compressed(
        os.path.join(HERE, f"case_{name}.npz"),
        meta=np.array([dogs, octaves, int(subpixel)] + list(meta or [0, 0, 0]), np.int64),
        level_names=np.array(sorted(levels)), level_sha=np.array([levels[k] for k in sorted(levels)]),
        n_candidates=np.int64(cand.size), cand_flags_sha=np.array(sha(cand["filtered"].astype(np.uint8))),
        cand_xy_sha=np.array(sha(np.stack([cand["x"], cand["y"], cand["octave"]], 1))),
        counts=np.array([run.points(s)[0].size for s in ("candidates", "after_sort1", "after_orient", "after_sort2", "final")], np.int64),
        kp_x=fin["x"], kp_y=fin["y"], kp_octave=fin["octave"], kp_index=fin["index"], kp_scale=fin["scale"],
        kp_orientation=fin["orientation"], descriptors=desc)
    print(name, "final", fin.size)


def make_digest_case(name, img, dogs, octaves, subpixel, meta, throwing_octaves=None):
    """Large cases (BASELINE config 5): only digests are committed, the arrays would be tens of MB."""
    out = {}
    if throwing_octaves is not None:   # the reference throws for this many octaves (App. B-14): keep its message
        run = O.OracleRun(img, dogs, throwing_octaves, subpixel=subpixel)
        assert run.status == 1
        out["throw_octaves"] = np.int64(throwing_octaves)
        out["throw_message"] = np.array(run.error)
        run.close()
    run = O.OracleRun(img, dogs, octaves, subpixel=subpixel)
    assert run.status == 0, run.error
    fin, desc = run.points("final")
    counts = np.array([run.points(s)[0].size for s in ("candidates", "after_sort1", "after_orient", "after_sort2", "final")], np.int64)
    np.savez_compressed(os.path.join(HERE, f"digest_{name}.npz"),
                        meta=np.array([dogs, octaves, int(subpixel)] + list(meta), np.int64), counts=counts,
                        kp_sha=np.array(sha(np.stack([fin["x"], fin["y"], fin["octave"], fin["index"]], 1).astype(np.uint16))),
                        orientation_sha=np.array(sha(fin["orientation"])), scale_sha=np.array(sha(fin["scale"])),
                        descriptors_sha=np.array(sha(desc)), first_kp=np.array([fin["x"][0], fin["y"][0], fin["octave"][0]], np.int64),
                        **out)
    print(name, "final", fin.size, "oracle seconds", run.seconds)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "truncation":
        # App. B-7: more than 65535 points survive the edge filter, `u16_t size` keeps count mod 65536
        make_digest_case("u16_truncation_4k", synth_frame(3840, 2160, 11), 3, 4, False, [3840, 2160, 11])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "config5":
        # BASELINE config 5: 3840x2160, subpixel, 6 octaves throws in the reference; 5 octaves runs
        make_digest_case("config5_4k", synth_frame(3840, 2160, 3), 3, 5, True, [3840, 2160, 3], throwing_octaves=6)
        return
    pgm = os.path.join(HERE, "parrot_r.pgm")
    if os.path.exists("/root/reference/example/parrot.jpg"):
        from PIL import Image
        r = np.asarray(Image.open("/root/reference/example/parrot.jpg").convert("RGB"))[:, :, 0]
        write_pgm(pgm, r)
    taps = {f"taps_{i}": O.gauss_taps(s)[1] for i, s in enumerate(SIGMAS)}
    luts = {f"lut_{a}_{b}": O.resize_index_map(a, b) for a, b in LUTS}
    par = np.array([O.lib().oracle_vertex_parabola(355, 0.0, 5, h, 15, 0.0) for h in (1.0, 37.5, 1234.567, 98765.4, 3.3e6)], np.float32)
    np.savez_compressed(os.path.join(HERE, "kats.npz"), sigmas=np.array(SIGMAS, np.float32), parabola=par, **taps, **luts)
    make_case("synth_96x80", synth_frame(96, 80, 3), 3, 2, meta=[96, 80, 3])
    make_case("synth_160x120", synth_frame(160, 120, 1), 3, 2, meta=[160, 120, 1])
    make_case("synth_200x150_sub", synth_frame(200, 150, 2), 3, 2, subpixel=True, meta=[200, 150, 2])
    make_case("parrot", read_pgm(pgm), 3, 4)


if __name__ == "__main__":
    main()
