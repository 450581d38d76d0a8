"""ctypes binding of the CPU oracle (oracle/).  TEST INFRASTRUCTURE: only tests/, smoke() and
bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.path.join(_ROOT, "oracle", "build", "liboracle_sift.so")


class OracleParams(C.Structure):
    _fields_ = [("dogs_per_epoch", C.c_uint16), ("octaves", C.c_uint16), ("sigma", C.c_float),
                ("k", C.c_float), ("subpixel", C.c_uint8)]


POINT_DTYPE = np.dtype([("scale", "<f4"), ("orientation", "<f4"), ("x", "<u2"), ("y", "<u2"),
                        ("octave", "<u2"), ("index", "<u2"), ("filtered", "<u4"),
                        ("cand_id", "<i4"), ("n_desc", "<i4")])
assert POINT_DTYPE.itemsize == 28

K_SQRT2 = float(np.float32(np.sqrt(2.0)))

_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        L.oracle_run.restype = C.c_void_p
        L.oracle_run.argtypes = [fp, C.c_int, C.c_int, C.POINTER(OracleParams), C.c_int, C.c_char_p, C.c_int]
        L.oracle_status.argtypes = [C.c_void_p]
        L.oracle_seconds.restype = C.c_double
        L.oracle_seconds.argtypes = [C.c_void_p]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_image_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_image_copy.argtypes = [C.c_void_p, fp]
        L.oracle_level_dims.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_level_copy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, fp]
        L.oracle_level_scale.restype = C.c_float
        L.oracle_level_scale.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oracle_points_count.argtypes = [C.c_void_p, C.c_int]
        L.oracle_points_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_write_result.argtypes = [C.c_void_p, C.c_char_p]
        L.oracle_gauss_taps.argtypes = [C.c_float, fp, C.c_int]
        L.oracle_convolve.argtypes = [fp, C.c_int, C.c_int, C.c_float, fp, C.c_char_p, C.c_int]
        L.oracle_resize_index_map.argtypes = [C.c_int, C.c_int, ip]
        L.oracle_resample.argtypes = [fp, C.c_int, C.c_int, C.c_float, C.c_int, fp, C.c_char_p, C.c_int]
        L.oracle_dog.argtypes = [fp, fp, C.c_int, C.c_int, fp]
        L.oracle_vertex_parabola.restype = C.c_float
        L.oracle_vertex_parabola.argtypes = [C.c_uint16, C.c_float, C.c_uint16, C.c_float, C.c_uint16, C.c_float]
        L.oracle_edge_filtered.argtypes = [fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_gradient.argtypes = [fp, C.c_int, C.c_int, fp, fp]
        L.oracle_inverse3.argtypes = [fp, fp]
        L.oracle_solve3.argtypes = [fp, fp, fp]
        L.oracle_sort_by_filter.argtypes = [np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), C.c_int, ip]
        L.oracle_atan2f.restype = C.c_float
        L.oracle_atan2f.argtypes = [C.c_float, C.c_float]
        L.oracle_f32_to_u16.restype = C.c_uint16
        L.oracle_f32_to_u16.argtypes = [C.c_float]
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


class OracleRun:
    """One Sift::calculate() of the oracle with every intermediate kept."""

    KINDS = {"gaussian": 0, "dog": 1, "magnitude": 2, "orientation": 3}
    STAGES = {"candidates": 0, "after_sort1": 1, "after_orient": 2, "after_sort2": 3, "final": 4}

    def __init__(self, img, dogs=3, octaves=4, sigma=1.6, k=K_SQRT2, subpixel=False, faithful=False):
        img = np.ascontiguousarray(img, dtype=np.float32)
        h, w = img.shape
        self.params = OracleParams(dogs, octaves, sigma, k, 1 if subpixel else 0)
        err = C.create_string_buffer(512)
        self._h = lib().oracle_run(img, w, h, C.byref(self.params), 1 if faithful else 0, err, 512)
        self.status = lib().oracle_status(self._h)
        self.error = err.value.decode()
        self.seconds = lib().oracle_seconds(self._h)
        self.dogs, self.octaves = dogs, octaves

    def close(self):
        if self._h:
            lib().oracle_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def image(self):
        w, h = C.c_int(), C.c_int()
        lib().oracle_image_dims(self._h, C.byref(w), C.byref(h))
        out = np.empty((h.value, w.value), np.float32)
        lib().oracle_image_copy(self._h, out)
        return out

    def level(self, kind, o, i):
        w, h = C.c_int(), C.c_int()
        if not lib().oracle_level_dims(self._h, self.KINDS[kind], o, i, C.byref(w), C.byref(h)) or w.value == 0:
            return None
        out = np.empty((h.value, w.value), np.float32)
        lib().oracle_level_copy(self._h, self.KINDS[kind], o, i, out)
        return out

    def scale(self, kind, o, i):
        return lib().oracle_level_scale(self._h, self.KINDS[kind], o, i)

    def write_result(self, path):
        """interstpoints.txt as the reference's main.cpp:78-89 writes it (C++ iostream formatting)."""
        if lib().oracle_write_result(self._h, os.fsencode(path)):
            raise OSError(f"cannot write {path}")

    def points(self, stage="final"):
        s = self.STAGES[stage]
        n = lib().oracle_points_count(self._h, s)
        pts = np.zeros(n, POINT_DTYPE)
        desc = np.zeros((n, 128), np.float32)
        if n:
            lib().oracle_points_copy(self._h, s, pts.ctypes.data, desc.ctypes.data)
        return pts, desc


def gauss_taps(sigma):
    buf = np.zeros(4096, np.float32)
    r = lib().oracle_gauss_taps(sigma, buf, buf.size)
    return r, buf[:2 * r + 1].copy()


def convolve(img, sigma):
    img = np.ascontiguousarray(img, np.float32)
    out = np.empty_like(img)
    err = C.create_string_buffer(512)
    st = lib().oracle_convolve(img, img.shape[1], img.shape[0], sigma, out, err, 512)
    if st:
        raise OracleError(st, err.value.decode())
    return out


def resize_index_map(wold, wnew):
    out = np.zeros(wnew, np.int32)
    lib().oracle_resize_index_map(wold, wnew, out)
    return out


def resample(img, sigma, mode):
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    shape = ((h + 1) // 2, (w + 1) // 2) if mode == 0 else (2 * h, 2 * w)
    out = np.empty(shape, np.float32)
    err = C.create_string_buffer(512)
    st = lib().oracle_resample(img, w, h, sigma, mode, out, err, 512)
    if st:
        raise OracleError(st, err.value.decode())
    return out


def sort_by_filter(flags):
    flags = np.ascontiguousarray(flags, np.uint8)
    perm = np.zeros(flags.size, np.int32)
    lib().oracle_sort_by_filter(flags, flags.size, perm)
    return perm
