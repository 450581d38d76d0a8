// Two blurs in one pass: g(0,0) = blur(input) and g(0,1) = blur(g(0,0)), the first two launches of every pyramid
// (/root/reference/sift.cpp:386-398: the first octave's first level is convolveWithGauss(img, sigma), its second level
// convolveWithGauss of that with gauss_scale[1] = sigma).  Run as two launches, g(0,0) is written to HBM and read straight
// back; here the second blur takes its rows from LDS while the first one is still walking down the image, and HBM sees
// one read of the input and one write of each level.
//
// Every WAVE is on its own, like the streaming blur (kernels_pyramid.hip): a strip of 256 columns (4 per lane), a chunk of
// rows, walked top to bottom.
//   stage 1: input row -> LDS row -> row pass -> sliding column sums A1 (2R+1 partial sums per column in registers);
//            each step completes one row of g(0,0), which goes to HBM and into a per-wave ring of 2R+1 rows in LDS;
//   stage 2: R rows behind stage 1, takes the rows of g(0,0) the second blur consumes IN THE REFERENCE'S ORDER (reflected
//            rows above the first / below the last row come out of the ring a second time), row pass, sliding column sums
//            A2, one row of g(0,1) per step.
// The outer RA columns each side of a strip are the second blur's halo (their g(0,0) is computed, their g(0,1) is not), so a
// strip delivers 256 - 2*RA columns; where a halo leaves the image its columns are filled in LDS by reflection from the
// columns the wave holds (input row and ring row alike), which gives the second blur g(0,0)[reflect(x)], bit for bit what
// the reference reads.  Sums are formed exactly as in the streaming blur: ascending from 0.0f, one rounding per multiply
// and per add.
#include <hip/hip_ext.h>

#include "common.h"
#include "blur_pair.h"   // blur_pair_kernel (rounds 5), blur_pair2_kernel (round 6)

#pragma clang fp contract(off)

namespace sift_hip {

// waves a launch is cut into (option "pair_waves" of the calling context; <= 0: this default).  The chip holds 2048 of this kernel's
// waves (two per SIMD); alone 2048 is fastest (180 us), beside the other batch's kernels 1536 - 1792 (profiles/r05_pair_ab.txt)
constexpr int kPairWaves = 1536;

template <int R>
static bool launch_pair_r(hipStream_t s, const float* in, float* g0, float* g1, int w, int h, int n, const float* taps,
                          int min_waves, int pair_waves, hipEvent_t ev_start, hipEvent_t ev_stop) {
    constexpr int SWU = pair_useful(R);
    constexpr int RI = pair_runin(R);
    if (w % 4 != 0 || w < SWU || h < 4 * R + 2) return false;
    if ((((uintptr_t)in | (uintptr_t)g0 | (uintptr_t)g1) & 15u) != 0) return false;
    const int strips = (w + SWU - 1) / SWU;
    int chunks = (pair_waves > 0 ? pair_waves : kPairWaves) / (n * strips);
    if (chunks < 1) chunks = 1;
    int chunk_h = (h + chunks - 1) / chunks;
    // short chunks pay stage 1's and stage 2's run-in (3R + RI rows) too often: batches of 4 - 8 frames of 1080p are no faster
    // this way than as two launches (profiles/r05_pair_ab.txt); the tests' forced mode (min_waves = 1) takes any chunk
    if (min_waves > 1 && chunk_h < 64) return false;
    if (chunk_h < 3 * RI) chunk_h = 3 * RI;
    if (chunk_h > h) chunk_h = h;
    chunks = (h + chunk_h - 1) / chunk_h;
    const int total = n * strips * chunks;
    if (total < min_waves) return false;
    hipExtLaunchKernelGGL((blur_pair_kernel<R>), dim3((unsigned)total), dim3(64), 0, s, ev_start, ev_stop, 0, in, g0, g1, w, h, strips,
                          chunks, chunk_h, total, taps);
    return true;
}

// g0 = blur(in), g1 = blur(g0), both with the same taps.  false: the shape is not one this kernel takes (the caller runs
// the two blurs one after the other).
bool launch_blur_pair(hipStream_t s, const float* in, float* g0, float* g1, int w, int h, int n, const float* taps,
                      int radius, int min_waves, int pair_waves, hipEvent_t ev_start, hipEvent_t ev_stop) {
    switch (radius) {
        case 3: return launch_pair_r<3>(s, in, g0, g1, w, h, n, taps, min_waves, pair_waves, ev_start, ev_stop);
        case 4: return launch_pair_r<4>(s, in, g0, g1, w, h, n, taps, min_waves, pair_waves, ev_start, ev_stop);
        case 5: return launch_pair_r<5>(s, in, g0, g1, w, h, n, taps, min_waves, pair_waves, ev_start, ev_stop);
        case 6: return launch_pair_r<6>(s, in, g0, g1, w, h, n, taps, min_waves, pair_waves, ev_start, ev_stop);
        default: return false;
    }
}

__global__ void tu_probe_pair_kernel() {}
void tu_touch_pair(hipStream_t s) { hipLaunchKernelGGL(tu_probe_pair_kernel, dim3(1), dim3(1), 0, s); }

}  // namespace sift_hip
