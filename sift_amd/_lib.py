"""ctypes binding of libsift_hip.so — the C ABI declared in include/sift_hip.h.

There is no fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIFT_HIP_LIBRARY: file name of another build in sift_amd/lib (tools/ use libsift_hip_diag.so, `make -C sift_amd/csrc diag`: the
# measurement options); read here, in the binding - the library itself reads no environment variable
LIB_PATH = os.path.join(_HERE, "lib", os.path.basename(os.environ.get("SIFT_HIP_LIBRARY", "libsift_hip.so")))

OK, EPRECONDITION, EASSERT, EINVAL, EHIP = 0, 1, 2, 3, 4

# every symbol include/sift_hip.h declares
SYMBOLS = [
    "sift_hip_create", "sift_hip_destroy", "sift_hip_set_option", "sift_hip_calculate_batch",
    "sift_hip_calculate_batch_device", "sift_hip_result_images", "sift_hip_result_status", "sift_hip_result_counts",
    "sift_hip_result_total", "sift_hip_result_copy", "sift_hip_result_device", "sift_hip_image_dims",
    "sift_hip_image_copy", "sift_hip_level_dims", "sift_hip_level_copy", "sift_hip_level_scale",
    "sift_hip_stage_count", "sift_hip_stage_copy", "sift_hip_gauss_taps", "sift_hip_convolve_with_gauss",
    "sift_hip_reduce_to_next_level", "sift_hip_increase_to_next_level", "sift_hip_dog", "sift_hip_gradient",
    "sift_hip_edge_responses", "sift_hip_vertex_parabola", "sift_hip_sort_by_filter", "sift_hip_cleanup_survivors", "sift_hip_profile_get", "sift_hip_profile_get_busy",
    "sift_hip_profile_reset", "sift_hip_profile_batches", "sift_hip_version", "sift_hip_gate_create", "sift_hip_gate_destroy", "sift_hip_set_gate",
    "sift_hip_result_sparse_size", "sift_hip_result_sparse_pack", "sift_hip_result_sparse_pack_async", "sift_hip_result_pack_wait", "sift_hip_sparse_unpack",
    "sift_hip_host_alloc", "sift_hip_host_free",
    "sift_hip_group_create", "sift_hip_group_destroy", "sift_hip_group_shards", "sift_hip_group_set_option", "sift_hip_group_calculate",
    "sift_hip_group_result_images", "sift_hip_group_result_status", "sift_hip_group_result_counts", "sift_hip_group_result_total",
    "sift_hip_group_result_copy", "sift_hip_group_result_device", "sift_hip_group_timing", "sift_hip_group_submit", "sift_hip_group_collect",
    "sift_hip_group_transport", "sift_hip_group_gather_exposed", "sift_hip_lock_wait_ms", "sift_hip_calculate_batch_u8", "sift_hip_calculate_batch_device_u8",
    "sift_hip_result_copy_sparse", "sift_hip_sparse_unpack_host",
    "sift_hip_image_info", "sift_hip_image_read_band0", "sift_hip_image_read_bgr8", "sift_hip_png_write_bgr8",
    "sift_hip_rotated_rect_points", "sift_hip_overlay_box", "sift_hip_overlay_draw",
]


class Params(C.Structure):
    _fields_ = [("dogs_per_epoch", C.c_uint16), ("octaves", C.c_uint16), ("sigma", C.c_float),
                ("k", C.c_float), ("subpixel", C.c_uint8), ("reserved", C.c_uint8 * 3)]


KEYPOINT_DTYPE = np.dtype([("scale", "<f4"), ("orientation", "<f4"), ("x", "<u2"), ("y", "<u2"),
                           ("octave", "<u2"), ("index", "<u2"), ("filtered", "u1"),
                           ("has_descriptor", "u1"), ("reserved", "<u2")])
assert KEYPOINT_DTYPE.itemsize == 20

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so.7 (and HSA runtime) next to
    libtorch; libsift_hip.so asks for the same SONAME and would otherwise pull in /opt/rocm's copy when it is
    loaded before torch, after which torch finds no GPU ("No HIP GPUs are available": the second HSA runtime
    cannot open the device).  Loading torch's copy first (without importing torch) makes the order irrelevant:
    the dynamic linker then resolves our dependency to the runtime that is already there."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


def load():
    """Load libsift_hip.so and declare prototypes.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the HIP path)")
    # contexts that run side by side (BatchPipeline) need hardware queues of their own; the HIP runtime reads this when it
    # initialises, the library itself never writes the environment
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    vp, ip, cs, ci = C.c_void_p, C.POINTER(C.c_int), C.c_char_p, C.c_int
    L.sift_hip_version.restype = C.c_char_p
    L.sift_hip_create.argtypes = [ci, C.POINTER(vp), cs, ci]
    L.sift_hip_destroy.argtypes = [vp]
    L.sift_hip_destroy.restype = None
    L.sift_hip_gate_create.argtypes = [ci, C.POINTER(vp)]
    L.sift_hip_gate_destroy.argtypes = [vp]
    L.sift_hip_gate_destroy.restype = None
    L.sift_hip_set_gate.argtypes = [vp, vp]
    L.sift_hip_result_sparse_size.argtypes = [vp, C.POINTER(C.c_int64), ip]
    L.sift_hip_result_sparse_pack.argtypes = [vp, vp, vp]
    L.sift_hip_result_sparse_pack_async.argtypes = [vp, vp, vp]
    L.sift_hip_result_pack_wait.argtypes = [vp]
    L.sift_hip_sparse_unpack.argtypes = [vp, vp, vp, C.c_int64, vp, vp]
    L.sift_hip_set_option.argtypes = [vp, cs, ci]
    L.sift_hip_calculate_batch.argtypes = [vp, fp, ci, ci, ci, C.POINTER(Params), cs, ci]
    L.sift_hip_calculate_batch_device.argtypes = [vp, vp, ci, ci, ci, C.POINTER(Params), cs, ci]
    L.sift_hip_result_images.argtypes = [vp]
    L.sift_hip_result_status.argtypes = [vp, i32p, ci]
    L.sift_hip_result_counts.argtypes = [vp, i32p, ci]
    L.sift_hip_result_total.argtypes = [vp]
    L.sift_hip_result_total.restype = C.c_int64
    L.sift_hip_result_copy.argtypes = [vp, vp, vp]
    L.sift_hip_result_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.sift_hip_image_dims.argtypes = [vp, ip, ip]
    L.sift_hip_image_copy.argtypes = [vp, ci, fp]
    L.sift_hip_level_dims.argtypes = [vp, ci, ci, ci, ip, ip]
    L.sift_hip_level_copy.argtypes = [vp, ci, ci, ci, ci, fp]
    L.sift_hip_level_scale.argtypes = [vp, ci, ci, ci]
    L.sift_hip_level_scale.restype = C.c_float
    L.sift_hip_stage_count.argtypes = [vp, ci, ci]
    L.sift_hip_stage_copy.argtypes = [vp, ci, ci, vp]
    L.sift_hip_gauss_taps.argtypes = [C.c_float, fp, ci]
    L.sift_hip_convolve_with_gauss.argtypes = [vp, fp, ci, ci, C.c_float, fp, cs, ci]
    L.sift_hip_reduce_to_next_level.argtypes = [vp, fp, ci, ci, C.c_float, fp, cs, ci]
    L.sift_hip_increase_to_next_level.argtypes = [vp, fp, ci, ci, C.c_float, fp, cs, ci]
    L.sift_hip_dog.argtypes = [vp, fp, fp, ci, ci, fp]
    L.sift_hip_gradient.argtypes = [vp, fp, ci, ci, fp, fp]
    L.sift_hip_edge_responses.argtypes = [vp, fp, fp, fp, ci, ci, u16p, u16p, ci, u8p]
    L.sift_hip_vertex_parabola.argtypes = [vp, u16p, fp, u16p, fp, u16p, fp, ci, fp]
    L.sift_hip_sort_by_filter.argtypes = [vp, u8p, ci, i32p]
    L.sift_hip_cleanup_survivors.argtypes = [vp, u8p, ci, i32p, C.POINTER(C.c_int32), ci]
    L.sift_hip_profile_get.argtypes = [vp, ci, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.sift_hip_profile_get_busy.argtypes = [vp, ci, C.POINTER(C.c_double)]
    L.sift_hip_profile_reset.argtypes = [vp]
    L.sift_hip_profile_batches.argtypes = [vp, C.POINTER(C.c_int64)]
    ll = C.c_longlong
    L.sift_hip_group_create.argtypes = [ip, ci, C.POINTER(vp), cs, ci]
    L.sift_hip_group_destroy.argtypes = [vp]
    L.sift_hip_group_destroy.restype = None
    L.sift_hip_group_shards.argtypes = [vp]
    L.sift_hip_group_set_option.argtypes = [vp, cs, ci]
    L.sift_hip_group_calculate.argtypes = [vp, fp, ci, ci, ci, C.POINTER(Params), cs, ci]
    L.sift_hip_calculate_batch_u8.argtypes = [vp, vp, ci, ci, ci, C.POINTER(Params), cs, ci]
    L.sift_hip_calculate_batch_device_u8.argtypes = [vp, vp, ci, ci, ci, C.POINTER(Params), cs, ci]
    L.sift_hip_result_copy_sparse.argtypes = [vp, vp, vp]
    L.sift_hip_sparse_unpack_host.argtypes = [vp, vp, C.c_int64, vp, vp, ci]
    L.sift_hip_group_submit.argtypes = [vp, fp, ci, ci, ci, C.POINTER(Params), cs, ci]
    L.sift_hip_group_collect.argtypes = [vp, cs, ci]
    L.sift_hip_group_transport.argtypes = [vp, cs, ci]
    L.sift_hip_group_gather_exposed.argtypes = [vp, C.POINTER(C.c_double)]
    L.sift_hip_lock_wait_ms.argtypes = [C.POINTER(C.c_double)]
    L.sift_hip_group_result_images.argtypes = [vp]
    L.sift_hip_group_result_status.argtypes = [vp, i32p, ci]
    L.sift_hip_group_result_counts.argtypes = [vp, i32p, ci]
    L.sift_hip_group_result_total.argtypes = [vp]
    L.sift_hip_group_result_total.restype = C.c_int64
    L.sift_hip_group_result_copy.argtypes = [vp, vp, vp]
    L.sift_hip_group_result_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.sift_hip_group_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.sift_hip_host_alloc.argtypes = [C.c_size_t]
    L.sift_hip_host_alloc.restype = vp
    L.sift_hip_host_free.argtypes = [vp]
    L.sift_hip_host_free.restype = None
    L.sift_hip_image_info.argtypes = [cs, ip, ip, ip, ip, cs, ci]
    L.sift_hip_image_read_band0.argtypes = [cs, fp, ll, cs, ci]
    L.sift_hip_image_read_bgr8.argtypes = [cs, u8p, ll, cs, ci]
    L.sift_hip_png_write_bgr8.argtypes = [cs, u8p, ci, ci, cs, ci]
    L.sift_hip_rotated_rect_points.argtypes = [C.c_float] * 5 + [fp]
    L.sift_hip_rotated_rect_points.restype = None
    L.sift_hip_overlay_box.argtypes = [vp, ci, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), ip, fp]
    L.sift_hip_overlay_box.restype = None
    L.sift_hip_overlay_draw.argtypes = [u8p, ci, ci, vp, ll, ci]
    _lib = L
    return L
