import hashlib
import os

import numpy as np

from sift_amd.synthetic import synth_frame

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["synth_96x80", "synth_160x120", "synth_200x150_sub", "parrot"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def read_pgm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P5"
        w, h = map(int, f.readline().split())
        assert int(f.readline()) == 255
        return np.frombuffer(f.read(), np.uint8).reshape(h, w).astype(np.float32)


def load_case(name):
    g = np.load(os.path.join(GOLDEN, f"case_{name}.npz"))
    dogs, octaves, subpixel, w, h, seed = (int(v) for v in g["meta"])
    img = read_pgm(os.path.join(GOLDEN, "parrot_r.pgm")) if name == "parrot" else synth_frame(w, h, seed)
    return g, img, dogs, octaves, bool(subpixel)
