"""The C-ABI library loads on a CPU-only machine and exports every symbol include/sift_hip.h
declares; without a GPU every compute entry fails loudly (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from sift_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_library_agree():
    hdr = open(os.path.join(ROOT, "include", "sift_hip.h")).read()
    declared = set(re.findall(r"\b(sift_hip_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    L = _lib.load()
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.sift_hip_version()


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sift_amd.sift import Context, HipError
    with pytest.raises(HipError):
        Context(0)


def test_host_glue_entry_points_without_gpu():
    """Tap tables and the cleanup permutation are host code: they must agree with the oracle."""
    import oracle_lib as O
    from sift_amd.sift import gauss_taps
    for sigma in (1.0, 1.6, 2.2627418, 3.2, 4.5254836, 6.4, 9.050967, 12.8, 18.101934):
        r, t = gauss_taps(sigma)
        r2, t2 = O.gauss_taps(sigma)
        assert r == r2 and t.tobytes() == t2.tobytes()
    # the streaming blur kernel shares the product tap[j] * x between the slots j and 2r - j of its column pass:
    # the tables must be symmetric bit for bit (they are: initGaussian evaluates x * x)
    for sigma in np.concatenate([np.linspace(0.2, 5.0, 97), [6.4, 9.050967, 12.8, 18.101934]]).astype(np.float32):
        r, t = gauss_taps(float(sigma))
        assert t.size == 2 * r + 1 and t.tobytes() == t[::-1].tobytes(), float(sigma)
    L = _lib.load()
    rng = np.random.default_rng(0)
    for n in (0, 1, 15, 16, 17, 33, 100, 1000, 4097, 70000):
        for p in (0.0, 0.1, 0.5, 0.9, 1.0):
            flags = (rng.random(n) < p).astype(np.uint8)
            perm = np.zeros(n, np.int32)
            assert L.sift_hip_sort_by_filter(None, flags, n, perm) == 0
            assert (perm == O.sort_by_filter(flags)).all(), (n, p)
