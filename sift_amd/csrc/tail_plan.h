// Argument block of pyramid_tail_kernel (kernels_tail.hip): the ops of the pyramid's tail in execution order.
#pragma once
#include <hip/hip_runtime.h>

namespace sift_hip {

constexpr int kMaxTailOps = 24;
constexpr int kTailLdsFloats = 39936;   // 156 KiB of the CU's 160 KiB: the ring of row-pass rows + the staging rows

struct TailOp {
    int kind;            // 2: level blur (+ DoG), 3: reduceToNextLevel (BlurOp::kind, context.cpp)
    int w, h;            // size of the level the blur runs on
    int wd, hd;          // size of what is stored (kind 3: the next octave's; else w, h)
    int radius;
    int tap_off;         // float offset of the op's taps in the context's tap table
    int lut_x, lut_y;    // kind 3: offsets of the destination -> source index maps in the context's map table
    int band;            // output rows per band (tail_band_rows)
    const float* src;    // level bases: the batch's images back to back
    float* dst;          // may be null (the top level of an octave is not kept: option lazy_top)
    float* dog;          // null for kind 3
};

struct TailPlan {
    int n_ops;
    TailOp op[kMaxTailOps];
};

int tail_band_rows(int w, int h, int radius);   // 0: the level does not fit the kernel
void launch_pyramid_tail(hipStream_t s, const TailPlan& plan, int n_images, const float* d_taps, const int* d_luts,
                         hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void tu_touch_tail(hipStream_t s);

}  // namespace sift_hip
