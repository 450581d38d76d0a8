"""Python mirror of the reference's `sift::Sift` / `sift::InterestPoint` API
(/root/reference/sift.hpp:17-78, interestpoint.hpp:13-63) on top of the C ABI (include/sift_hip.h).

Same constructor arguments, same meaning, same error behaviour: Vigra precondition failures the
reference would throw surface as `PreconditionViolation`, its assert()s as `AssertionError`.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib

K_SQRT2 = float(np.float32(np.sqrt(2.0)))

KINDS = {"gaussian": 0, "dog": 1, "magnitude": 2, "orientation": 3}
STAGES = {"candidates": 0, "after_sort1": 1, "after_orient": 2, "after_sort2": 3, "final": 4}


class PreconditionViolation(RuntimeError):
    """vigra::PreconditionViolation analogue; str() carries Vigra's message text."""


class HipError(RuntimeError):
    pass


@dataclass
class InterestPoint:  # interestpoint.hpp:13-63
    scale: float
    octave: int
    index: int
    filtered: bool
    loc: tuple
    orientation: float
    descriptors: list = field(default_factory=list)


def _raise(rc, err):
    msg = err.value.decode(errors="replace")
    if rc == _lib.EPRECONDITION:
        raise PreconditionViolation(msg)
    if rc == _lib.EASSERT:
        raise AssertionError(msg)
    if rc == _lib.EINVAL:
        raise ValueError(msg or "sift_hip: invalid argument")
    raise HipError(msg or f"sift_hip error {rc}")


def pinned_array(shape, dtype=np.float32) -> np.ndarray:
    """numpy array in page-locked host memory (sift_hip_host_alloc): the copy engines move it at the PCIe rate in one
    asynchronous copy, where ordinary memory goes through the library's staging buffers (include/sift_hip.h).  The memory
    is released when the array (and every view of it) is gone."""
    import weakref
    L = _lib.load()
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    p = L.sift_hip_host_alloc(max(n, 1))
    if not p:
        raise MemoryError(f"sift_hip_host_alloc({n}) failed")
    buf = (C.c_char * max(n, 1)).from_address(p)
    weakref.finalize(buf, L.sift_hip_host_free, p)
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class Gate:
    """sift_hip_gate: orders the phases of batches that run on different contexts of one GPU (include/sift_hip.h)."""

    def __init__(self, device: int = 0):
        self._L = _lib.load()
        self._h = C.c_void_p()
        if self._L.sift_hip_gate_create(device, C.byref(self._h)):
            self._h = C.c_void_p()
            raise HipError(f"sift_hip_gate_create({device}) failed")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.sift_hip_gate_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One sift_hip_ctx (one GPU, one stream)."""

    def __init__(self, device: int = 0):
        self._L = _lib.load()
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self._L.sift_hip_create(device, C.byref(self._h), err, 512)
        if rc:
            self._h = C.c_void_p()
            _raise(rc, err)
        self.device = device

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.sift_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_gate(self, gate: "Gate | None"):
        self._gate = gate    # keeps the Gate object alive as long as this context points at it
        if self._L.sift_hip_set_gate(self._h, gate._h if gate is not None else None):
            raise ValueError("gate and context are on different devices, or the gate already joins its maximum of four contexts")

    def set_option(self, name: str, value: int):
        if self._L.sift_hip_set_option(self._h, name.encode(), int(value)):
            raise ValueError(f"unknown option {name}")

    # ---- Sift::calculate ---------------------------------------------------------------------
    def calculate_batch(self, imgs, params, raise_on_error=True):
        """Frames from host memory.  A uint8 array takes the 8-bit entry point (a quarter of the bytes cross the link, the GPU
        widens them to the floats vigra::importImage would have produced); anything else is handed over as float32."""
        imgs = np.asarray(imgs)
        u8 = imgs.dtype == np.uint8
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8 if u8 else np.float32)
        if imgs.ndim == 2:
            imgs = imgs[None]
        n, h, w = imgs.shape
        err = C.create_string_buffer(512)
        if u8:
            rc = self._L.sift_hip_calculate_batch_u8(self._h, imgs.ctypes.data, n, w, h, C.byref(params), err, 512)
        else:
            rc = self._L.sift_hip_calculate_batch(self._h, imgs.reshape(-1), n, w, h, C.byref(params), err, 512)
        if rc and raise_on_error:
            _raise(rc, err)
        return rc, err.value.decode(errors="replace")

    def calculate_batch_device(self, dev_ptr: int, n: int, w: int, h: int, params, raise_on_error=True):
        err = C.create_string_buffer(512)
        rc = self._L.sift_hip_calculate_batch_device(self._h, C.c_void_p(dev_ptr), n, w, h, C.byref(params), err, 512)
        if rc and raise_on_error:
            _raise(rc, err)
        return rc, err.value.decode(errors="replace")

    def n_images(self) -> int:
        """Images of the batch whose results the context holds (the library's count, not the caller's)."""
        n = int(self._L.sift_hip_result_images(self._h))
        if n < 0:
            raise HipError("no result")
        return n

    def counts(self):
        out = np.zeros(self.n_images(), np.int32)
        if self._L.sift_hip_result_counts(self._h, out, out.size):
            raise HipError("no result")
        return out

    def status(self):
        out = np.zeros(self.n_images(), np.int32)
        if self._L.sift_hip_result_status(self._h, out, out.size):
            raise HipError("no result")
        return out

    def total(self) -> int:
        return int(self._L.sift_hip_result_total(self._h))

    def results(self, kp_out=None, desc_out=None):
        """(keypoints, descriptors [total, 128]).  `kp_out` / `desc_out`: caller arrays to fill (e.g. `pinned_array`s with
        room for at least total() entries); views of their first total() entries are returned."""
        t = max(self.total(), 0)
        kp = np.zeros(t, _lib.KEYPOINT_DTYPE) if kp_out is None else kp_out.reshape(-1)[:t]
        desc = np.zeros((t, 128), np.float32) if desc_out is None else desc_out.reshape(-1, 128)[:t]
        if kp.size < t or desc.shape[0] < t:
            raise ValueError("result arrays too small")
        if t > 0 and self._L.sift_hip_result_copy(self._h, kp.ctypes.data, desc.ctypes.data):
            raise HipError("sift_hip_result_copy failed")
        return kp, desc

    def results_sparse(self, rec_out=None, val_out=None):
        """The results in the sparse wire format, in host memory: (records uint8 [total, 34] = 20-byte keypoint record + 112
        presence bits, values float32 [n] = the descriptor floats that are not +0.0f) - what crosses the link is ~200 instead of
        532 bytes per keypoint.  `unpack_sparse_host` turns them back into (keypoints, descriptors).  Falls back to None when the
        results hold a value the format would lose (see sift_hip_result_sparse_size)."""
        t = max(self.total(), 0)
        nnz, lossless = C.c_int64(), C.c_int()
        if self._L.sift_hip_result_sparse_size(self._h, C.byref(nnz), C.byref(lossless)):
            raise HipError("sift_hip_result_sparse_size failed")
        if not lossless.value:
            return None
        rec = np.zeros((t, 34), np.uint8) if rec_out is None else rec_out.reshape(-1)[:t * 34].reshape(t, 34)
        val = np.zeros(nnz.value, np.float32) if val_out is None else val_out.reshape(-1)[:nnz.value]
        if val.size < nnz.value or rec.shape[0] < t:
            raise ValueError("result arrays too small")
        if t > 0 and self._L.sift_hip_result_copy_sparse(self._h, rec.ctypes.data, val.ctypes.data):
            raise HipError("sift_hip_result_copy_sparse failed")
        return rec, val

    def result_device_ptrs(self):
        a, b = C.c_void_p(), C.c_void_p()
        if self._L.sift_hip_result_device(self._h, C.byref(a), C.byref(b)):
            raise HipError("no result")
        return a.value or 0, b.value or 0

    def sparse_size(self, require_lossless: bool = True) -> int:
        """Number of descriptor floats the sparse wire format carries for the current results (include/sift_hip.h).
        Raises ValueError when the format cannot carry them (some bin 7 is not +0.0f): use the "full" wire then."""
        n, ok = C.c_int64(), C.c_int()
        if self._L.sift_hip_result_sparse_size(self._h, C.byref(n), C.byref(ok)):
            raise HipError("sift_hip_result_sparse_size failed")
        if require_lossless and not ok.value:
            raise ValueError("the sparse wire format would lose a bin 7 that is not +0.0f: send the full descriptors")
        return int(n.value)

    def sparse_pack(self, dev_records: int, dev_values: int, wait: bool = True):
        """Write total*34 record bytes and sparse_size() floats to the device addresses given.  wait=False: the pack is only
        queued (side stream) and this context's next batch may be started at once; `pack_wait()` - from any thread - returns
        when the lists are complete."""
        f = self._L.sift_hip_result_sparse_pack if wait else self._L.sift_hip_result_sparse_pack_async
        if f(self._h, C.c_void_p(dev_records), C.c_void_p(dev_values)):
            raise HipError("sift_hip_result_sparse_pack failed")

    def pack_wait(self):
        if self._L.sift_hip_result_pack_wait(self._h):
            raise HipError("sift_hip_result_pack_wait failed")

    def sparse_unpack(self, dev_records: int, dev_values: int, n_keypoints: int, dev_keypoints: int, dev_descriptors: int):
        """34-byte records + set floats (device addresses, from any context's sparse_pack) -> n_keypoints 20-byte keypoint
        records and n_keypoints * 128 floats at the device addresses given, on this context's GPU."""
        if self._L.sift_hip_sparse_unpack(self._h, C.c_void_p(dev_records), C.c_void_p(dev_values), int(n_keypoints),
                                          C.c_void_p(dev_keypoints), C.c_void_p(dev_descriptors)):
            raise HipError("sift_hip_sparse_unpack failed")

    def image(self, image: int = 0):
        w, h = C.c_int(), C.c_int()
        self._L.sift_hip_image_dims(self._h, C.byref(w), C.byref(h))
        out = np.empty((h.value, w.value), np.float32)
        if self._L.sift_hip_image_copy(self._h, image, out):
            return None
        return out

    # ---- inspection ------------------------------------------------------------------------------
    def level(self, kind: str, octave: int, level: int, image: int = 0):
        w, h = C.c_int(), C.c_int()
        if self._L.sift_hip_level_dims(self._h, KINDS[kind], octave, level, C.byref(w), C.byref(h)) or w.value == 0:
            return None
        out = np.empty((h.value, w.value), np.float32)
        if self._L.sift_hip_level_copy(self._h, image, KINDS[kind], octave, level, out):
            raise HipError("sift_hip_level_copy failed")
        return out

    def level_scale(self, kind: str, octave: int, level: int) -> float:
        return float(self._L.sift_hip_level_scale(self._h, KINDS[kind], octave, level))

    def stage(self, name: str, image: int = 0):
        s = STAGES[name]
        n = self._L.sift_hip_stage_count(self._h, image, s)
        if n < 0:
            raise HipError("no such stage")
        out = np.zeros(n, _lib.KEYPOINT_DTYPE)
        if n and self._L.sift_hip_stage_copy(self._h, image, s, out.ctypes.data):
            raise HipError("sift_hip_stage_copy failed")
        return out

    # ---- sift::alg operators -------------------------------------------------------------------
    def _op(self, fn, img, sigma, shape):
        img = np.ascontiguousarray(img, np.float32)
        h, w = img.shape
        out = np.empty(shape, np.float32)
        err = C.create_string_buffer(512)
        rc = fn(self._h, img, w, h, sigma, out, err, 512)
        if rc:
            _raise(rc, err)
        return out

    def convolve_with_gauss(self, img, sigma):
        return self._op(self._L.sift_hip_convolve_with_gauss, img, sigma, np.shape(img))

    def reduce_to_next_level(self, img, sigma):
        h, w = np.shape(img)
        return self._op(self._L.sift_hip_reduce_to_next_level, img, sigma, ((h + 1) // 2, (w + 1) // 2))

    def increase_to_next_level(self, img, sigma):
        h, w = np.shape(img)
        return self._op(self._L.sift_hip_increase_to_next_level, img, sigma, (2 * h, 2 * w))

    def dog(self, lower, higher):
        lower = np.ascontiguousarray(lower, np.float32)
        higher = np.ascontiguousarray(higher, np.float32)
        out = np.empty_like(lower)
        if self._L.sift_hip_dog(self._h, lower, higher, lower.shape[1], lower.shape[0], out):
            raise HipError("sift_hip_dog failed")
        return out

    def gradient(self, img):
        img = np.ascontiguousarray(img, np.float32)
        mag, ori = np.empty_like(img), np.empty_like(img)
        if self._L.sift_hip_gradient(self._h, img, img.shape[1], img.shape[0], mag, ori):
            raise HipError("sift_hip_gradient failed")
        return mag, ori

    def edge_responses(self, d0, d1, d2, xs, ys):
        d0, d1, d2 = (np.ascontiguousarray(a, np.float32) for a in (d0, d1, d2))
        xs = np.ascontiguousarray(xs, np.uint16)
        ys = np.ascontiguousarray(ys, np.uint16)
        flags = np.zeros(xs.size, np.uint8)
        if self._L.sift_hip_edge_responses(self._h, d0, d1, d2, d0.shape[1], d0.shape[0], xs, ys, xs.size, flags):
            raise HipError("sift_hip_edge_responses failed")
        return flags

    def vertex_parabola(self, lnx, lny, px, py, rnx, rny):
        a = [np.ascontiguousarray(v, np.uint16) for v in (lnx, px, rnx)]
        b = [np.ascontiguousarray(v, np.float32) for v in (lny, py, rny)]
        out = np.zeros(a[0].size, np.float32)
        if self._L.sift_hip_vertex_parabola(self._h, a[0], b[0], a[1], b[1], a[2], b[2], a[0].size, out):
            raise HipError("sift_hip_vertex_parabola failed")
        return out

    def sort_by_filter(self, flags):
        flags = np.ascontiguousarray(flags, np.uint8)
        perm = np.zeros(flags.size, np.int32)
        if self._L.sift_hip_sort_by_filter(self._h, flags, flags.size, perm):
            raise HipError("sift_hip_sort_by_filter failed")
        return perm

    def profile(self, which: int):
        ms, n, by = C.c_double(), C.c_int64(), C.c_double()
        self._L.sift_hip_profile_get(self._h, which, C.byref(ms), C.byref(n), C.byref(by))
        return ms.value, n.value, by.value

    def profile_busy_ms(self, which: int) -> float:
        """time during which at least one launch of the class ran (overlapping launches counted once)"""
        ms = C.c_double()
        self._L.sift_hip_profile_get_busy(self._h, which, C.byref(ms))
        return ms.value

    def profile_batches(self) -> int:
        """Batches whose launches carried timing events since the last profile_reset()."""
        n = C.c_int64()
        self._L.sift_hip_profile_batches(self._h, C.byref(n))
        return int(n.value)

    def profile_reset(self):
        self._L.sift_hip_profile_reset(self._h)


class Group:
    """sift_hip_group: one batch block-sharded over several GPUs of this node from one process (one context and host thread
    per entry of `devices`; a device may be listed more than once), keypoint lists gathered device-to-device on devices[0] in
    global image order (include/sift_hip.h)."""

    def __init__(self, devices):
        self._L = _lib.load()
        self._h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        err = C.create_string_buffer(512)
        rc = self._L.sift_hip_group_create(devs, len(devices), C.byref(self._h), err, 512)
        if rc:
            self._h = C.c_void_p()
            _raise(rc, err)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.sift_hip_group_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int):
        if self._L.sift_hip_group_set_option(self._h, name.encode(), int(value)):
            raise ValueError(f"unknown option {name}")

    def calculate_batch(self, imgs, params, raise_on_error=True):
        imgs = np.ascontiguousarray(imgs, dtype=np.float32)
        n, h, w = imgs.shape
        err = C.create_string_buffer(512)
        rc = self._L.sift_hip_group_calculate(self._h, imgs.reshape(-1), n, w, h, C.byref(params), err, 512)
        if rc and raise_on_error:
            _raise(rc, err)
        return rc, err.value.decode(errors="replace")

    def submit(self, imgs, params):
        """Start a batch and return at once (at most two in flight); `collect()` waits for the oldest one.  The gather of a batch
        runs under the kernels of the next."""
        imgs = np.ascontiguousarray(imgs, dtype=np.float32)
        n, h, w = imgs.shape
        err = C.create_string_buffer(512)
        rc = self._L.sift_hip_group_submit(self._h, imgs.reshape(-1), n, w, h, C.byref(params), err, 512)
        if rc:
            _raise(rc, err)
        self._inflight = getattr(self, "_inflight", []) + [imgs]    # the frames stay alive until their batch is collected

    def collect(self, raise_on_error=True):
        err = C.create_string_buffer(512)
        rc = self._L.sift_hip_group_collect(self._h, err, 512)
        if getattr(self, "_inflight", None):
            self._inflight.pop(0)
        if rc and raise_on_error:
            _raise(rc, err)
        return rc, err.value.decode(errors="replace")

    def transport(self):
        """(1 if the gather runs over RCCL else 0, why)"""
        text = C.create_string_buffer(256)
        return int(self._L.sift_hip_group_transport(self._h, text, 256)), text.value.decode(errors="replace")

    def gather_exposed_ms(self):
        v = C.c_double()
        if self._L.sift_hip_group_gather_exposed(self._h, C.byref(v)):
            raise HipError("no batch collected")
        return v.value

    def counts(self):
        out = np.zeros(max(self._L.sift_hip_group_result_images(self._h), 0), np.int32)
        if self._L.sift_hip_group_result_counts(self._h, out, out.size):
            raise HipError("no result")
        return out

    def status(self):
        out = np.zeros(max(self._L.sift_hip_group_result_images(self._h), 0), np.int32)
        if self._L.sift_hip_group_result_status(self._h, out, out.size):
            raise HipError("no result")
        return out

    def total(self) -> int:
        return int(self._L.sift_hip_group_result_total(self._h))

    def results(self):
        t = max(self.total(), 0)
        kp = np.zeros(t, _lib.KEYPOINT_DTYPE)
        desc = np.zeros((t, 128), np.float32)
        if t and self._L.sift_hip_group_result_copy(self._h, kp.ctypes.data, desc.ctypes.data):
            raise HipError("sift_hip_group_result_copy failed")
        return kp, desc

    def timing(self):
        """(ms in the shards' calculate calls, ms in the gather, bytes that crossed devices) of the last batch"""
        a, b, n = C.c_double(), C.c_double(), C.c_int64()
        if self._L.sift_hip_group_timing(self._h, C.byref(a), C.byref(b), C.byref(n)):
            raise HipError("no result")
        return a.value, b.value, n.value


def lock_wait_ms() -> float:
    """Time this process's host threads have spent waiting for the library's per-device launch locks (sift_hip_lock_wait_ms)."""
    v = C.c_double()
    if _lib.load().sift_hip_lock_wait_ms(C.byref(v)):
        raise HipError("sift_hip_lock_wait_ms failed")
    return v.value


def unpack_sparse_host(rec, val, kp_out=None, desc_out=None, threads: int = 8):
    """Sparse wire format in host memory -> (keypoints, descriptors [n, 128]), bit for bit what Context.results returns
    (sift_hip_sparse_unpack_host: plain host code, `threads` threads)."""
    L = _lib.load()
    rec = np.ascontiguousarray(rec, np.uint8).reshape(-1, 34)
    val = np.ascontiguousarray(val, np.float32)
    n = rec.shape[0]
    kp = np.zeros(n, _lib.KEYPOINT_DTYPE) if kp_out is None else kp_out.reshape(-1)[:n]
    desc = np.empty((n, 128), np.float32) if desc_out is None else desc_out.reshape(-1, 128)[:n]
    if L.sift_hip_sparse_unpack_host(rec.ctypes.data, val.ctypes.data, n, kp.ctypes.data, desc.ctypes.data, int(threads)):
        raise HipError("sift_hip_sparse_unpack_host failed")
    return kp, desc


def gauss_taps(sigma: float):
    L = _lib.load()
    buf = np.zeros(8192, np.float32)
    r = L.sift_hip_gauss_taps(sigma, buf, buf.size)
    if r < 0:
        raise PreconditionViolation("Kernel1D::initGaussian(): Standard deviation must be >= 0.")
    return r, buf[:2 * r + 1].copy()


class Sift:
    """sift::Sift (sift.hpp:17-78): Sift(dogsPerEpoch=3, octaves=3, sigma=1.6, k=sqrt(2), subpixel=False)."""

    def __init__(self, dogsPerEpoch: int = 3, octaves: int = 3, sigma: float = 1.6, k: float = K_SQRT2,
                 subpixel: bool = False, device: int = 0, context: Context | None = None):
        self.subpixel = bool(subpixel)
        self._params = _lib.Params(dogsPerEpoch, octaves, sigma, k, 1 if subpixel else 0)
        self.ctx = context or Context(device)

    @property
    def params(self):
        return self._params

    def calculate(self, img: np.ndarray):
        """Returns (points, img): the InterestPoint list and the image the reference leaves in the
        caller's array (the 2x upsampled one when subpixel, sift.cpp:20-21)."""
        img = np.asarray(img)
        img = np.ascontiguousarray(img, np.uint8 if img.dtype == np.uint8 else np.float32)   # uint8: widened on the GPU
        try:
            self.ctx.calculate_batch(img[None], self._params)
        finally:
            self.image = self.ctx.image(0) if self.subpixel else img
        sparse = self.ctx.results_sparse() if self.ctx.total() > 0 else None    # ~200 instead of 532 bytes per keypoint over the link
        kp, desc = unpack_sparse_host(*sparse, threads=1) if sparse is not None else self.ctx.results()
        pts = [InterestPoint(float(k["scale"]), int(k["octave"]), int(k["index"]), bool(k["filtered"]),
                             (int(k["x"]), int(k["y"])), float(k["orientation"]),
                             desc[i].tolist() if k["has_descriptor"] else [])
               for i, k in enumerate(kp)]
        return pts

    def calculate_batch(self, imgs: np.ndarray):
        """Batch of independent frames -> (counts, keypoints structured array, descriptors [N,128])."""
        self.ctx.calculate_batch(imgs, self._params)
        kp, desc = self.ctx.results()
        return self.ctx.counts(), kp, desc
