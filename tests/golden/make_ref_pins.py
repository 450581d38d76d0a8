#!/usr/bin/env python3
"""Pins for the CPU oracle, produced by the reference's OWN prebuilt binary.

    python tests/golden/make_ref_pins.py                                   (in the build container, where /root/reference is mounted)
    python tests/golden/make_ref_pins.py bench_frame | config2 | truncation | configs | config5   (the long-running pins, one fixture each)

/root/reference/bin/arch_x64/sift cannot be started here (Vigra, OpenCV, Boost are DT_NEEDED and absent) and the
sources cannot be rebuilt for the same reason, but `Sift::calculate` and the `sift::alg` functions inside it only need
libc / libm / libstdc++: oracle/refexec maps the executable in its own process and calls them (see its header).  This
script runs it on the inputs below and stores WHAT THE REFERENCE RETURNED — point records, exception texts, SHA-256 of
descriptors, of every Gaussian and DoG level and of the final (mutated) gradient maps, a few small arrays in full — in tests/golden/refpin.npz.  The fixture holds
data only; tests/test_ref_pins.py checks the oracle against it anywhere (no reference needed), and against a live run
of the binary where it is present.
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import read_pgm, sha  # noqa: E402
from sift_amd.synthetic import blob_frame, synth_frame  # noqa: E402

REF_BIN = "/root/reference/bin/arch_x64/sift"
REFEXEC = os.path.join(ROOT, "oracle", "_ref", "refexec")
K_SQRT2 = float(np.float32(np.sqrt(np.float32(2.0))))

POINT_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("octave", "<u2"), ("index", "<u2"), ("filtered", "<u2"), ("pad", "<u2"),
                        ("scale", "<f4"), ("orientation", "<f4"), ("n_desc", "<u4")])

# name -> (image spec, dogs, octaves, subpixel).  image spec: ("synth", w, h, seed) | ("parrot",) | ("const", w, h, value)
CALC_CASES = {
    "synth_200x160_3x2": (("synth", 200, 160, 2), 3, 2, 0),
    "synth_320x250_3x3": (("synth", 320, 250, 4), 3, 3, 0),
    "synth_322x251_3x3": (("synth", 322, 251, 12), 3, 3, 0),
    "synth_640x480_3x3": (("synth", 640, 480, 14), 3, 3, 0),
    "synth_768x576_4x2": (("synth", 768, 576, 15), 4, 2, 0),
    "synth_200x150_3x3_subpixel": (("synth", 200, 150, 7), 3, 3, 1),
    "parrot_3x4": (("parrot",), 3, 4, 0),                         # BASELINE.json configs[0]
    "const_160x120_3x2": (("const", 160, 120, 7.0), 3, 2, 0),
    "throws_160x120_3x4": (("synth", 160, 120, 1), 3, 4, 0),      # App. B-13
    "throws_160x120_5x3": (("synth", 160, 120, 1), 5, 3, 0),
    "throws_512x384_4x3": (("synth", 512, 384, 3), 4, 3, 0),      # App. B-14 (dead 16x16 blur) if it throws
    "synth_1000x760_3x3": (("synth", 1000, 760, 21), 3, 3, 0),     # 122 k candidates, 7.4 k survivors: the unstable sort's order at scale
    # (App. B-7, more than 65535 SURVIVORS, needs ~8 Mpx: hours in the reference's O(K x N) descriptor stage - not pinned)
}
BLUR_SIGMAS = [0.3, 1.0, 1.6, 2.0, 2.2627418, 3.2, 4.5254836, 6.4]
BLUR_IMAGES = [("synth", 200, 150, 3), ("synth", 67, 131, 4), ("synth", 64, 64, 5)]
PARABOLAS = [(3, 1.5, 4, 2.5, 5, 1.0), (0, 0.25, 1, 0.75, 2, 0.5), (34, 12.0, 35, 12.0, 0, 3.0), (10, 1.0, 11, 1.0, 12, 1.0)]


def make_image(spec):
    if spec[0] == "synth":
        return synth_frame(spec[1], spec[2], spec[3])
    if spec[0] == "const":
        return np.full((spec[2], spec[1]), spec[3], np.float32)
    return read_pgm(os.path.join(HERE, "parrot_r.pgm"))


def refexec(*args):
    r = subprocess.run([REFEXEC, REF_BIN] + [str(a) for a in args], capture_output=True, text=True)
    return r.returncode, r.stdout, r.stderr


def ref_calculate(img, dogs, octaves, subpixel, tmp):
    """-> dict with what the reference returned (or its exception text)."""
    h, w = img.shape
    src = os.path.join(tmp, "in.f32")
    img.astype(np.float32).tofile(src)
    out = os.path.join(tmp, "out")
    rc, so, se = refexec("calculate", src, w, h, dogs, octaves, repr(float(np.float32(1.6))), repr(K_SQRT2), subpixel, out)
    if rc == 5:
        assert so.startswith("EXCEPTION "), so
        return {"exception": so[len("EXCEPTION "):].rstrip("\n") if so.endswith("\n\n") is False else so[len("EXCEPTION "):]}
    assert rc == 0, (rc, so, se)
    pts = np.fromfile(out + ".points", np.uint8).view(POINT_DTYPE).reshape(-1)
    desc = np.fromfile(out + ".desc", np.float32)
    meta = np.fromfile(out + ".levels_meta", np.int64)
    levels = np.fromfile(out + ".levels", np.float32)
    dims = np.fromfile(out + ".image_dims", np.int64)
    mw, mh = int(meta[0]), int(meta[1])
    per = meta[2:].reshape(mw * mh, 3)
    level_sha, off = [], 0
    for lw, lh, _ in per:
        level_sha.append(hashlib.sha256(levels[off:off + lw * lh].tobytes()).hexdigest())
        off += int(lw * lh)
    res = {"points": pts, "desc": desc, "levels_wh": (mw, mh), "level_dims": per[:, :2].copy(), "level_scale_bits": per[:, 2].astype(np.uint32),
           "level_sha": level_sha, "image_dims": dims, "levels": levels}
    # gradient maps the object keeps, in their FINAL state (the descriptor stage adds to them in place)
    for tag in ("mag", "ori"):
        mm = np.fromfile(out + f".{tag}_meta", np.int64)
        px = np.fromfile(out + f".{tag}", np.float32)
        shas, off = [], 0
        for lw, lh in mm[2:].reshape(-1, 2):
            shas.append(hashlib.sha256(px[off:off + lw * lh].tobytes()).hexdigest() if lw * lh else "")
            off += int(lw * lh)
        res[f"{tag}_sha"] = shas
    # the DoG pyramid: Sift::_createDOGs called on its own
    rc, so, se = refexec("dogs", src, w, h, dogs, octaves, repr(float(np.float32(1.6))), repr(K_SQRT2), subpixel, out)
    assert rc == 0, (rc, so, se)
    dm = np.fromfile(out + ".dogs_meta", np.int64)
    dl = np.fromfile(out + ".dogs", np.float32)
    dper = dm[2:].reshape(int(dm[0]) * int(dm[1]), 3)
    dsha, off = [], 0
    for lw, lh, _ in dper:
        dsha.append(hashlib.sha256(dl[off:off + lw * lh].tobytes()).hexdigest())
        off += int(lw * lh)
    res.update({"dogs_wh": (int(dm[0]), int(dm[1])), "dog_dims": dper[:, :2].copy(), "dog_scale_bits": dper[:, 2].astype(np.uint32), "dog_sha": dsha})
    return res


def main():
    assert os.path.exists(REF_BIN), "the reference is not mounted here"
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/refexec"])
    store = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (spec, dogs, octaves, sub) in CALC_CASES.items():
            img = make_image(spec)
            r = ref_calculate(img, dogs, octaves, sub, tmp)
            store[f"calc/{name}/params"] = np.array([dogs, octaves, sub, img.shape[1], img.shape[0]], np.int64)
            store[f"calc/{name}/image_sha"] = np.array(sha(img))
            if "exception" in r:
                store[f"calc/{name}/exception"] = np.array(r["exception"])
                print(f"{name}: EXCEPTION {r['exception']!r}")
                continue
            store[f"calc/{name}/points"] = r["points"]
            store[f"calc/{name}/desc_sha"] = np.array(hashlib.sha256(r["desc"].tobytes()).hexdigest())
            store[f"calc/{name}/levels_wh"] = np.array(r["levels_wh"], np.int64)
            store[f"calc/{name}/level_dims"] = r["level_dims"]
            store[f"calc/{name}/level_scale_bits"] = r["level_scale_bits"]
            store[f"calc/{name}/level_sha"] = np.array(r["level_sha"])
            store[f"calc/{name}/image_dims"] = r["image_dims"]
            store[f"calc/{name}/mag_sha"] = np.array(r["mag_sha"])
            store[f"calc/{name}/ori_sha"] = np.array(r["ori_sha"])
            store[f"calc/{name}/dogs_wh"] = np.array(r["dogs_wh"], np.int64)
            store[f"calc/{name}/dog_dims"] = r["dog_dims"]
            store[f"calc/{name}/dog_scale_bits"] = r["dog_scale_bits"]
            store[f"calc/{name}/dog_sha"] = np.array(r["dog_sha"])
            if name == "synth_200x160_3x2":      # one small case in full
                store[f"calc/{name}/desc"] = r["desc"]
                store[f"calc/{name}/levels"] = r["levels"]
            print(f"{name}: {r['points'].size} points, {int((r['points']['n_desc'] == 128).sum())} with descriptors")
        for spec in BLUR_IMAGES:
            img = make_image(spec)
            h, w = img.shape
            src = os.path.join(tmp, "in.f32")
            img.tofile(src)
            for sigma in BLUR_SIGMAS:
                key = f"blur/{spec[1]}x{spec[2]}_s{spec[3]}/{sigma!r}"
                rc, so, se = refexec("blur", src, w, h, repr(float(np.float32(sigma))), os.path.join(tmp, "o.f32"))
                if rc == 5:
                    store[key + "/exception"] = np.array(so[len("EXCEPTION "):])
                    continue
                assert rc == 0, (rc, so, se)
                store[key + "/sha"] = np.array(sha(np.fromfile(os.path.join(tmp, "o.f32"), np.float32)))
            for op in ("reduce", "increase"):
                rc, so, se = refexec(op, src, w, h, repr(float(np.float32(1.6))), os.path.join(tmp, "r"))
                assert rc == 0, (rc, so, se)
                store[f"{op}/{spec[1]}x{spec[2]}_s{spec[3]}/dims"] = np.fromfile(os.path.join(tmp, "r.dims"), np.int64)
                store[f"{op}/{spec[1]}x{spec[2]}_s{spec[3]}/sha"] = np.array(sha(np.fromfile(os.path.join(tmp, "r.f32"), np.float32)))
        a, b = synth_frame(200, 150, 3), synth_frame(200, 150, 9)
        a.tofile(os.path.join(tmp, "a.f32"))
        b.tofile(os.path.join(tmp, "b.f32"))
        rc, so, se = refexec("dog", os.path.join(tmp, "a.f32"), os.path.join(tmp, "b.f32"), 200, 150, os.path.join(tmp, "d.f32"))
        assert rc == 0, (rc, so, se)
        store["dog/200x150_s3_s9/sha"] = np.array(sha(np.fromfile(os.path.join(tmp, "d.f32"), np.float32)))
        par = []
        for p in PARABOLAS:
            rc, so, se = refexec("parabola", *p)
            assert rc == 0, (rc, so, se)
            par.append(int(so.strip(), 16))
        store["parabola/args"] = np.array(PARABOLAS, np.float64)
        store["parabola/bits"] = np.array(par, np.uint32)
    dst = os.path.join(HERE, os.environ.get("REFPIN_OUT", "refpin.npz"))
    np.savez_compressed(dst, **store)
    print("wrote", dst, os.path.getsize(dst), "bytes")


def long_running(which):
    """The pins that take the reference tens of minutes to hours (it copies three DoG images per candidate and re-blurs a
    level per keypoint); each goes to its own fixture.
        bench_frame   frame 1 of the bench workload, 1920x1080, 3 DoGs x 4 octaves          (51 minutes)
        config2       BASELINE.json configs[1] exactly as written: 640x480 seed 1, 4 octaves x 3 DoGs   (about a minute)
        truncation    blob_frame 1024x1088 seed 5: 66260 survivors of the first cleanup, `u16_t size` keeps 724 (App. B-7); the
                      reference copies three DoG images per candidate (2.7 TB of copies for the 220 716 candidates)
        configs       BASELINE.json configs[2] exactly as written (throws after 14 s; configs[4] as written would run for days)
        config5       configs[4]'s failure mode (the dead 16x16 blur of an octave-5 keypoint) on a 1792x1792 frame, 6 octaves x 3 DoGs"""
    import time
    assert os.path.exists(REF_BIN), "the reference is not mounted here"
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/refexec"])
    with tempfile.TemporaryDirectory() as tmp:
        if which == "bench_frame":
            img = synth_frame(1920, 1080, 1)
            t0 = time.time()
            r = ref_calculate(img, 3, 4, 0, tmp)
            store = {"params": np.array([3, 4, 0, 1920, 1080], np.int64), "image_sha": np.array(sha(img)), "seconds": np.array(time.time() - t0),
                     "points": r["points"], "desc_sha": np.array(hashlib.sha256(r["desc"].tobytes()).hexdigest()),
                     "levels_wh": np.array(r["levels_wh"], np.int64), "level_dims": r["level_dims"],
                     "level_scale_bits": r["level_scale_bits"], "level_sha": np.array(r["level_sha"])}
            np.savez_compressed(os.path.join(HERE, "refpin_bench_frame.npz"), **store)
        elif which == "config2":
            # BASELINE.json configs[1] exactly as written: one 640x480 greyscale frame (seed 1: CASES[1] of tests/test_gpu_parity.py),
            # 4 octaves x 3 DoGs - octave 3 is 80x60 with radii up to 27
            img = synth_frame(640, 480, 1)
            t0 = time.time()
            r = ref_calculate(img, 3, 4, 0, tmp)
            store = {"params": np.array([3, 4, 0, 640, 480, 1], np.int64), "image_sha": np.array(sha(img)), "seconds": np.array(time.time() - t0),
                     "points": r["points"], "desc_sha": np.array(hashlib.sha256(r["desc"].tobytes()).hexdigest()),
                     "levels_wh": np.array(r["levels_wh"], np.int64), "level_dims": r["level_dims"],
                     "level_scale_bits": r["level_scale_bits"], "level_sha": np.array(r["level_sha"]),
                     "dogs_wh": np.array(r["dogs_wh"], np.int64), "dog_dims": r["dog_dims"], "dog_scale_bits": r["dog_scale_bits"],
                     "dog_sha": np.array(r["dog_sha"]), "mag_sha": np.array(r["mag_sha"]), "ori_sha": np.array(r["ori_sha"])}
            np.savez_compressed(os.path.join(HERE, "refpin_config2.npz"), **store)
            print("config2:", r["points"].size, "points in", float(store["seconds"]), "s")
        elif which == "truncation":
            w, h, seed = 1024, 1088, 5
            img = blob_frame(w, h, seed)
            src, out = os.path.join(tmp, "in.f32"), os.path.join(tmp, "out")
            img.tofile(src)
            t0 = time.time()
            rc, so, se = refexec("calculate", src, w, h, 3, 4, repr(float(np.float32(1.6))), repr(K_SQRT2), 0, out)
            store = {"params": np.array([3, 4, 0, w, h, seed], np.int64), "image_sha": np.array(sha(img)), "seconds": np.array(time.time() - t0),
                     "rc": np.array(rc), "stdout": np.array(so)}
            if rc == 0:
                store["points"] = np.fromfile(out + ".points", np.uint8).view(POINT_DTYPE).reshape(-1)
                store["desc"] = np.fromfile(out + ".desc", np.float32)
            np.savez_compressed(os.path.join(HERE, "refpin_u16_truncation.npz"), **store)
        elif which in ("configs", "config5"):
            # configs[4] as written (4K, subpixel, 6 octaves) would keep the reference busy for days before it reaches its throw -
            # the dead 16x16 blur of an octave-5 keypoint (sift.cpp:184, App. B-14).  The SAME failure mode on a frame the binary
            # can finish: 1792x1792 (octave 5 is 56x56: larger than the pyramid's largest radius, 54), 6 octaves x 3 DoGs, no
            # subpixel - kept in the same fixture, beside config 3
            cases = {"config3_as_written": (1920, 1080, 3, 5, 4, 1)} if which == "configs" else {"config5_octave5_1792": (1792, 1792, 5, 3, 6, 0)}
            dst = os.path.join(HERE, "refpin_configs_as_written.npz")
            store = dict(np.load(dst)) if os.path.exists(dst) else {}
            for name, (w, h, seed, dogs, octaves, sub) in cases.items():
                img = synth_frame(w, h, seed)
                src, out = os.path.join(tmp, "in.f32"), os.path.join(tmp, "out")
                img.tofile(src)
                t0 = time.time()
                rc, so, se = refexec("calculate", src, w, h, dogs, octaves, repr(float(np.float32(1.6))), repr(K_SQRT2), sub, out)
                store[name + "/params"] = np.array([dogs, octaves, sub, w, h, seed], np.int64)
                store[name + "/image_sha"] = np.array(sha(img))
                store[name + "/rc"] = np.array(rc)
                store[name + "/stdout"] = np.array(so)
                store[name + "/seconds"] = np.array(time.time() - t0)
            np.savez_compressed(dst, **store)
        else:
            raise SystemExit("bench_frame | config2 | truncation | configs | config5")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        long_running(sys.argv[1])
    else:
        main()
