"""CPU build of the scalar math the HIP kernels inline (fdlibm atan2f restatement, register
3x3 Householder QR) against libm and the oracle's Vigra-style generic restatement."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hm():
    path = os.path.join(ROOT, "sift_amd", "lib", "libsift_hostmath.so")
    # (always through make: a no-op when the library is newer than the headers it is built from)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sift_amd", "csrc"), "../lib/libsift_hostmath.so"])
    H = C.CDLL(path)
    fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    H.hostmath_atan2f_array.argtypes = [fp, fp, C.c_int, fp]
    H.hostmath_atan2f_sel_array.argtypes = [fp, fp, C.c_int, fp]
    H.hostmath_atan2f_sel_array.restype = C.c_int
    H.hostmath_inverse3.argtypes = [fp, fp]
    H.hostmath_solve3.argtypes = [fp, fp, fp]
    H.hostmath_vertex_parabola.restype = C.c_float
    H.hostmath_vertex_parabola.argtypes = [C.c_uint16, C.c_float, C.c_uint16, C.c_float, C.c_uint16, C.c_float]
    H.hostmath_hist8_bin_mismatches.restype = C.c_longlong
    H.hostmath_hist8_bin_mismatches.argtypes = [C.c_ulonglong, C.c_ulonglong, C.POINTER(C.c_uint)]
    H.hostmath_magnitude_array.argtypes = [fp, fp, C.c_longlong, fp]
    H.hostmath_magnitude_array.restype = C.c_longlong
    H.hostmath_div_mismatches.argtypes = [fp, fp, C.c_longlong]
    H.hostmath_div_mismatches.restype = C.c_longlong
    return H


def midpoint_pairs(rng, k, dx_steps_below=0):
    """(dx, dy) whose sqrt(dx^2 + dy^2) lies within ~2^-47 (relative) of the MIDPOINT of two neighbouring floats - where
    (float)sqrt(double) is decided by the last bits of the double square root: dx the float below the midpoint m (or a few
    floats lower), dy = sqrt(m^2 - dx^2) rounded to float."""
    base = (rng.random(k) * 2.0 ** rng.integers(-10, 10, k)).astype(np.float32)
    nxt = np.nextafter(base, np.float32(np.inf))
    m = (base.astype(np.float64) + nxt.astype(np.float64)) / 2
    dx = base.copy()
    for _ in range(dx_steps_below):
        dx = np.nextafter(dx, np.float32(0))
    dy = np.sqrt(np.maximum(m * m - dx.astype(np.float64) ** 2, 0)).astype(np.float32)
    return dx, dy, m


def test_gradient_magnitude_is_the_double_square_root_rounded_to_float(hm):
    """alg::gradientMagnitude (/root/reference/algorithms.cpp:108-111) is (float)sqrt((double)dx*dx + (double)dy*dy).  The
    gradient kernels take ONE Newton round on a 24-bit reciprocal-square-root estimate and fall back to the correctly rounded
    routine where a float rounding boundary lies within the round's error bound (sift_amd/csrc/grad_math.h); the host build
    of that function (its estimate truncated to 24 bits: the device's worst case) against numpy's correctly rounded sqrt."""
    rng = np.random.default_rng(11)
    n = 4_000_000
    dx = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 3, n)).astype(np.float32)
    dy = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 3, n)).astype(np.float32)
    dx[:1000] = 0.0                       # flat rows / columns, and dx = dy = 0
    dy[500:1500] = 0.0
    dx[2000:3000] = np.round(dx[2000:3000] * 8)   # integer differences: exact squares (3, 4 -> 5)
    dy[2000:3000] = np.round(dy[2000:3000] * 8)
    out = np.empty(n, np.float32)
    hm.hostmath_magnitude_array(dx, dy, n, out)
    want = np.sqrt(dx.astype(np.float64) ** 2 + dy.astype(np.float64) ** 2).astype(np.float32)
    assert (out.view(np.uint32) == want.view(np.uint32)).all()
    k = 1_000_000
    for steps in (0, 1):
        dxa, dya, m = midpoint_pairs(rng, k, steps)
        for bump in (0, 1, -1):           # dy one float up / down: results decided just above / just below the midpoint
            dyb = (dya.view(np.int32) + bump).view(np.float32)
            s = dxa.astype(np.float64) ** 2 + dyb.astype(np.float64) ** 2
            want = np.sqrt(s).astype(np.float32)
            near = np.abs(np.sqrt(s) - m) / m < 2.0 ** -44
            assert near.mean() > 0.2, (steps, bump, near.mean())
            hm.hostmath_magnitude_array(dxa, dyb, k, out[:k])
            assert (out[:k].view(np.uint32) == want.view(np.uint32)).all(), (steps, bump)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 3.4e38, 1e-30, 1.0], np.float32)
    ys, xs = np.meshgrid(special, special)
    ys, xs = np.ascontiguousarray(ys.ravel()), np.ascontiguousarray(xs.ravel())
    hm.hostmath_magnitude_array(xs, ys, xs.size, out[:xs.size])
    with np.errstate(all="ignore"):
        want = np.sqrt(xs.astype(np.float64) ** 2 + ys.astype(np.float64) ** 2).astype(np.float32)
    g = out[:xs.size]
    assert ((g.view(np.uint32) == want.view(np.uint32)) | (np.isnan(g) & np.isnan(want))).all()


def test_division_without_exponent_scaling_is_the_ieee_division(hm):
    """fdlibm_atan2f.h: div_in_range - the compiler's own division sequence without v_div_scale / v_div_fixup - for operands in
    [2^-60, 2^60]: the quotient `a / b` gives, bit for bit (what atanf's argument reduction divides is covered separately by
    test_atan2f_matches_libm, which runs the whole restatement against glibc)."""
    rng = np.random.default_rng(12)
    n = 10_000_000
    a = (rng.standard_normal(n) * 10.0 ** rng.integers(-15, 15, n)).astype(np.float32)
    b = (rng.standard_normal(n) * 10.0 ** rng.integers(-15, 15, n)).astype(np.float32)
    a[:100000] = rng.integers(-2550, 2550, 100000).astype(np.float32)      # exact quotients, zeros
    b[:100000] = rng.integers(1, 256, 100000).astype(np.float32)
    ok = (np.abs(b) > 2.0 ** -60) & (np.abs(b) < 2.0 ** 60) & ((a == 0) | ((np.abs(a) > 2.0 ** -60) & (np.abs(a) < 2.0 ** 60)))
    a, b = np.ascontiguousarray(a[ok]), np.ascontiguousarray(b[ok])
    assert a.size > n // 2
    assert hm.hostmath_div_mismatches(a, b, a.size) == 0


def test_hist8_bin_without_division_all_inputs(hm):
    """The descriptor kernel's histogram bin `(u16_t)floor(orientation / 45) % 7` (algorithms.cpp:143-145) computed with a
    reciprocal multiply and one exact-residual correction instead of the division: identical for ALL 2^32 float inputs
    (8 threads, ~40 s)."""
    from concurrent.futures import ThreadPoolExecutor
    parts, total = 64, 1 << 32
    step = total // parts

    def run(i):
        ex = C.c_uint(0)
        return hm.hostmath_hist8_bin_mismatches(i * step, step, C.byref(ex)), ex.value
    with ThreadPoolExecutor(8) as pool:
        res = list(pool.map(run, range(parts)))
    bad = [(n, hex(e)) for n, e in res if n]
    assert not bad, bad


def test_atan2f_matches_libm(hm):
    rng = np.random.default_rng(1)
    n = 2_000_000
    y = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    x = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    # image-like differences too
    y[::2] = (rng.integers(-2550, 2550, n // 2) / rng.integers(1, 64, n // 2)).astype(np.float32)
    x[::2] = (rng.integers(-2550, 2550, n // 2) / rng.integers(1, 64, n // 2)).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.4e38], np.float32)
    ys, xs = np.meshgrid(special, special)
    y = np.ascontiguousarray(np.concatenate([y, ys.ravel()]))
    x = np.ascontiguousarray(np.concatenate([x, xs.ravel()]))
    got = np.empty_like(y)
    hm.hostmath_atan2f_array(y, x, y.size, got)
    want = np.array([O.lib().oracle_atan2f(float(a), float(b)) for a, b in zip(y[-100:], x[-100:])], np.float32)
    assert (got[-100:].view(np.uint32) == want.view(np.uint32))[~np.isnan(want)].all()
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    ok = np.isnan(ref) | (np.abs(got - ref) <= 4e-7 * np.maximum(1.0, np.abs(ref)))
    assert ok.all()
    # bit-for-bit against glibc's atan2f on a 200k sample through the oracle library
    idx = rng.integers(0, y.size, 200_000)
    want = np.array([O.lib().oracle_atan2f(float(a), float(b)) for a, b in zip(y[idx], x[idx])], np.float32)
    g = got[idx]
    same = (g.view(np.uint32) == want.view(np.uint32)) | (np.isnan(g) & np.isnan(want))
    assert same.all()
    # the branch-free common path the gradient kernel takes (+ fallback) is the same function, bit for bit
    got2 = np.empty_like(y)
    ncommon = hm.hostmath_atan2f_sel_array(y, x, y.size, got2)
    assert ncommon > y.size // 3
    same = (got2.view(np.uint32) == got.view(np.uint32)) | (np.isnan(got2) & np.isnan(got))
    assert same.all()


def test_linalg3_matches_oracle(hm):
    rng = np.random.default_rng(0)
    L = O.lib()
    rankdef = 0
    for t in range(20000):
        kind = t % 5
        a = (rng.standard_normal(9) * 10.0 ** rng.integers(-3, 3)).astype(np.float32)
        if kind == 1:
            a[6:9] = 0
        if kind == 2:
            a[3:6] = a[0:3] * 2
        if kind == 3:
            a = np.round(a).astype(np.float32)
        if kind == 4:
            m = a.reshape(3, 3)
            a = (m + m.T).astype(np.float32).reshape(-1).copy()
        b = rng.standard_normal(3).astype(np.float32)
        r1, r2 = np.zeros(9, np.float32), np.zeros(9, np.float32)
        o1, o2 = L.oracle_inverse3(a, r1), hm.hostmath_inverse3(a, r2)
        assert o1 == o2 and (not o1 or r1.tobytes() == r2.tobytes())
        s1, s2 = np.zeros(3, np.float32), np.zeros(3, np.float32)
        q1, q2 = L.oracle_solve3(a, b, s1), hm.hostmath_solve3(a, b, s2)
        rankdef += not q1
        assert q1 == q2 and s1.tobytes() == s2.tobytes()
    assert rankdef > 1000  # the minimum-norm path is exercised


def test_vertex_parabola_kat(hm):
    # (355,0),(5,h0),(15,0) -> rank 2, ~177.4913 for any h0 > 0
    for h0 in (1.0, 37.5, 1234.567, 98765.4):
        v = hm.hostmath_vertex_parabola(355, 0.0, 5, h0, 15, 0.0)
        assert v == O.lib().oracle_vertex_parabola(355, 0.0, 5, h0, 15, 0.0)
        assert abs(v - 177.4913) < 1e-3
    assert np.isnan(hm.hostmath_vertex_parabola(355, 0.0, 5, 0.0, 15, 0.0))
    assert hm.hostmath_vertex_parabola(355, 0.0, 5, 37.5, 15, 0.0) == np.float32(177.49134826660156)


def test_sparse_unpack_host_restores_records_and_descriptors():
    """sift_hip_sparse_unpack_host (plain host code of the C ABI: the receiving side of sift_hip_result_copy_sparse) against a
    numpy construction of the wire format: 34-byte records = 20-byte keypoint + 112 presence bits (bit cell*7+bin, LSB first),
    then the floats whose bit pattern is not +0.0f in ascending position; bin 7 of every cell comes back as +0.0f; any number
    of threads gives the same bytes; -0.0f and NaN payloads survive."""
    import ctypes as C
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sift_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(11)
    n = 5003
    desc = rng.random((n, 16, 8)).astype(np.float32)
    desc[rng.random((n, 16, 8)) < 0.62] = 0.0
    desc[:, :, 7] = 0.0
    desc[3, 2, 1] = -0.0                       # not +0.0f: travels
    desc[4, 0, 0] = np.float32("nan")
    desc[5] = 0.0                              # a keypoint without a single set float
    kp = np.zeros(n, _lib.KEYPOINT_DTYPE)
    kp["x"] = rng.integers(0, 1920, n)
    kp["y"] = rng.integers(0, 1080, n)
    kp["scale"] = rng.random(n).astype(np.float32)
    kp["has_descriptor"] = 1
    d7 = np.ascontiguousarray(desc[:, :, :7]).reshape(n, 112)
    present = d7.view(np.uint32) != 0
    rec = np.ascontiguousarray(np.concatenate([kp.view(np.uint8).reshape(n, 20), np.packbits(present, axis=1, bitorder="little")], axis=1))
    val = np.ascontiguousarray(d7[present])
    assert rec.shape == (n, 34)
    want = desc.reshape(n, 128)
    for threads in (0, 1, 3, 8, 64):
        k2 = np.zeros(n, _lib.KEYPOINT_DTYPE)
        d2 = np.full((n, 128), 7.0, np.float32)
        assert L.sift_hip_sparse_unpack_host(rec.ctypes.data, val.ctypes.data, n, k2.ctypes.data, d2.ctypes.data, threads) == 0
        assert k2.tobytes() == kp.tobytes() and d2.tobytes() == want.tobytes(), threads
    # records only / descriptors only / nothing at all
    k2 = np.zeros(n, _lib.KEYPOINT_DTYPE)
    assert L.sift_hip_sparse_unpack_host(rec.ctypes.data, val.ctypes.data, n, k2.ctypes.data, None, 2) == 0 and k2.tobytes() == kp.tobytes()
    d2 = np.empty((n, 128), np.float32)
    assert L.sift_hip_sparse_unpack_host(rec.ctypes.data, val.ctypes.data, n, None, d2.ctypes.data, 2) == 0 and d2.tobytes() == want.tobytes()
    assert L.sift_hip_sparse_unpack_host(None, None, 0, None, None, 1) == 0


def test_fmod_360_of_the_gradient_orientation_is_one_exact_float_subtraction():
    """alg::gradientOrientation (/root/reference/algorithms.cpp:113-116) returns (float)fmod((double)(atan2f(dy, dx) + 360.f), 360.).
    The kernel (kernels_orient.hip: gradient_pixel) computes s = r + 360.f and `s >= 360.f ? s - 360.f : s` in float.  EVERY float
    the sum can be - all of [360 - 4, 360 + 4], a superset of 360 +- pi - gives the same bits both ways."""
    lo = np.float32(356.0).view(np.uint32)
    hi = np.float32(364.0).view(np.uint32)
    s = np.arange(int(lo), int(hi) + 1, dtype=np.uint32).view(np.float32)
    want = np.fmod(s.astype(np.float64), 360.0).astype(np.float32)
    got = np.where(s >= np.float32(360.0), s - np.float32(360.0), s).astype(np.float32)
    assert s.size > 250000
    assert got.tobytes() == want.tobytes()
