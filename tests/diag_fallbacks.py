"""The library's FALLBACK paths, forced onto ordinary inputs, against the CPU oracle.

libsift_hip.so takes these paths by itself when a shape does not fit its default kernels (rows that are not 16-byte
aligned, radii beyond 32, index maps without a parity split, an introsort that hits its depth limit); here they are
forced through option names that exist only in sift_amd/lib/libsift_hip_diag.so - the SHIPPED kernels and host objects
with context.cpp alone compiled -DSIFT_HIP_DIAG (`make -C sift_amd/csrc`, part of `all`).

Not collected by `pytest tests/` (the file name does not match): tests/test_gpu_parity.py::
test_forced_fallback_paths_in_the_diag_library runs it in a process of its own with SIFT_HIP_LIBRARY=libsift_hip_diag.so
(sift_amd/_lib.py reads that name at import), so that the test session proper only ever maps the shipped library.
By hand:  SIFT_HIP_LIBRARY=libsift_hip_diag.so python -m pytest tests/diag_fallbacks.py -m gpu -q
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from sift_amd import _lib
from sift_amd.synthetic import synth_frame
from test_gpu_parity import CASES, CONVOLVE_SIGMAS, REDUCE_CASES, compare_run, convolve_cases

pytestmark = pytest.mark.gpu


def test_this_is_the_diag_library(ctx):
    assert os.path.basename(_lib.LIB_PATH) == "libsift_hip_diag.so", _lib.LIB_PATH
    assert ctx._L.sift_hip_set_option(ctx._h, b"fused_blur", 1) == _lib.OK
    assert ctx._L.sift_hip_set_option(ctx._h, b"desc_dbg", 0) == _lib.EINVAL     # kernel ablations: `make ablate` only


class forced:
    def __init__(self, ctx, **options):
        self.ctx, self.options = ctx, options

    def __enter__(self):
        for k, v in self.options.items():
            self.ctx.set_option(k, v)

    def __exit__(self, *exc):
        defaults = {"stream_min_waves": 0}
        for k in self.options:
            self.ctx.set_option(k, defaults.get(k, 1))


def test_two_pass_blur_operator(ctx):
    """separableConvolveX into a temporary, then separableConvolveY (algorithms.cpp:15-21) for every radius"""
    with forced(ctx, fused_blur=0):
        for sigma in CONVOLVE_SIGMAS:
            convolve_cases(ctx, sigma, "fused_blur=0")


def test_two_pass_blur_pipeline(ctx, report_dir):
    with forced(ctx, fused_blur=0):
        compare_run(ctx, synth_frame(200, 160, 2), 3, 2, False, "two-pass blur 200x160", report_dir)


def test_separate_scan_and_edge_filter(ctx, report_dir):
    """mask kernel + thread-per-candidate edge filter over DoG levels the pyramid wrote (what rows that are not 16-byte
    aligned get)"""
    with forced(ctx, fused_edge=0):
        compare_run(ctx, synth_frame(320, 250, 4), 3, 3, False, "separate scan / edge filter 320x250", report_dir)


def test_separate_blur_and_decimation(ctx, report_dir):
    """reduceToNextLevel as blur -> temporary -> resampling kernel"""
    with forced(ctx, fused_reduce=0):
        compare_run(ctx, synth_frame(1024, 512, 8), 3, 3, False, "separate blur / decimation 1024x512", report_dir, batch_of=4)


@pytest.mark.parametrize("case", [REDUCE_CASES[0], REDUCE_CASES[1], REDUCE_CASES[3]], ids=["640x480", "1000x600", "1280x360"])
def test_streaming_decimating_blur(ctx, report_dir, case):
    """reduceToNextLevel as the streaming blur that computes every pixel and stores the kept quarter"""
    name, w, h, seed, octaves, frames = case
    with forced(ctx, stream_min_waves=1, reduce_kept=0):
        compare_run(ctx, synth_frame(w, h, seed), 3, octaves, False, name + " [reduce_kept = 0]", report_dir, batch_of=frames)


@pytest.mark.parametrize("case", [CASES[1], CASES[2], CASES[4]], ids=[CASES[1][0], CASES[2][0], CASES[4][0]])
@pytest.mark.parametrize("streaming", [0, 1], ids=["tile", "streaming"])
def test_dog_levels_written_by_the_pyramid(ctx, report_dir, case, streaming):
    """Every blur launch writes its DoG level (rounds 1 - 4's form; what a plan with a scanned octave whose rows are not
    16-byte aligned gets), the fused scan reads three DoG levels: levels, stage lists, descriptors."""
    name, w, h, seed, dogs, octaves, subpixel = case
    with forced(ctx, dog_in_extrema=0, stream_min_waves=streaming):
        rep = compare_run(ctx, synth_frame(w, h, seed), dogs, octaves, subpixel, name + " [dog_in_extrema = 0]", report_dir, batch_of=2)
        assert rep["final"] > 0


def test_host_glue_path(ctx, report_dir):
    """flags down, libstdc++'s std::sort on the host, lists up (sift.cpp:37-54)"""
    with forced(ctx, gpu_cleanup=0):
        compare_run(ctx, synth_frame(333, 257, 9), 3, 3, False, "host-glue path 333x257", report_dir)


@pytest.mark.parametrize("option", ["gate_schedule=0", "pyramid_side=0", "dog_in_extrema=0"])
def test_other_gate_schedules_leave_the_results_alone(ctx, option):
    """The other order of the phase gate (sift_amd/csrc/phase_gate.h: schedule 0 keeps the pyramids alone on the chip) and the
    pyramid without its side stream (every launch on one stream): same results."""
    option, value = option.split("=")
    from sift_amd.pipeline import BatchPipeline
    params = _lib.Params(3, 3, 1.6, O.K_SQRT2, 0)
    batches = [np.stack([synth_frame(480, 360, 200 + 3 * b + i) for i in range(3)]) for b in range(5)]
    got = []
    with BatchPipeline(0, depth=2, options={option: int(value)}) as pipe:
        tickets = []
        for b in batches + [None, None]:
            if b is not None:
                tickets.append(pipe.submit(b, params))
            if len(tickets) == 2 or (b is None and tickets):
                t = tickets.pop(0)
                c = t.result()
                got.append((c.counts().copy(),) + tuple(a.copy() for a in c.results()))
                t.release()
    assert len(got) == len(batches)
    for b, (counts, kp, desc) in zip(batches, got):
        ctx.calculate_batch(b, params)
        wkp, wdesc = ctx.results()
        assert counts.tolist() == ctx.counts().tolist() and kp.tobytes() == wkp.tobytes() and desc.tobytes() == wdesc.tobytes()
