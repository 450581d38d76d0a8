// Shared declarations between the HIP kernels and the host-side context of libsift_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sift_hip.h"

// Host threads and the HIP runtime.  Several contexts are driven by one host thread each (sift_hip_group's shard threads,
// BatchPipeline's workers).  On this runtime (HIP 7.2) kernel launches of one thread crashed - SEGV at address 0 a few frames
// below hipLaunchKernel, once a kernel that started with garbage arguments (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION) -
// while another thread had the runtime copy device memory to device memory (hipMemcpyAsync: a blit kernel the runtime launches
// itself): 3 - 8 % of the runs of examples/sift_multi_gpu.cpp (tools/example_loop.sh, tools/probe/segv_bt.c for the
// backtraces).  What was tried, in this order: every translation unit's device code built before any thread starts
// (tu_touch_*: kept, it removes the first-launch cost from the first batch); all launches of the library under one lock; also
// the allocations, stream / event creation, event records and the runtime's copies under that lock (all kept: the lock is held
// for the call only, a batch makes ~60 such calls of a few microseconds); no spin-wait; no null stream; 4 / 24 hardware queues -
// none of it changed the rate.  Replacing the runtime's device-to-device copies by kernels of this library (group.cpp) did:
// 1 crash in 200 runs, and none in 150 runs of examples/sift_pipeline.cpp, which never used such copies.
// Round 5, what ended it: the crashing instruction sits behind hipLaunchKernel's lookup of the kernel's HOST STUB in the runtime's
// table of registered functions, made on every launch; the library now makes that lookup once per (device, kernel) under the
// device's lock - hipGetFuncBySymbol - and launches the function object it got (hipExtModuleLaunchKernel: launch_cache.h).  Same
// box, blocks of 100 runs of examples/sift_multi_gpu.cpp alternately: 0 crashes in 300 against 13 in 300 for a build that differs
// only in -DSIFT_HIP_STATIC_LAUNCH (tools/launch_ab.sh, profiles/r05_launch_ab_soak.txt).  The locks below stay: they order the
// runtime's own calls per device and cost ~60 uncontended acquisitions per batch.
#include "launch_guard.h"   // one lock per device since round 4, waiting time accounted
namespace sift_hip {
// ... and so are the calls that create or destroy what a launch touches (device and pinned memory, streams, events): the
// crashes went on, always below hipLaunchKernel, until hipMalloc / hipFree / hipStreamCreate / hipEventCreate of one thread could
// no longer run beside a launch of another (tools/example_loop.sh: 9 of 120 runs before).
using ApiGuard = LaunchGuard;
}
// ... and the calls that put the runtime's OWN kernels and markers on a stream (copies and fills are blit kernels here; event
// records and stream waits are marker packets): the guard lives for the call (a temporary in a comma expression), never for a wait.
#define hipSetDevice(...) ((hipError_t)::sift_hip::set_device_tracked(__VA_ARGS__))   // the launch locks are per device: launch_guard.h
#define hipMemcpyAsync(...) (::sift_hip::LaunchGuard{}, (hipMemcpyAsync)(__VA_ARGS__))
#define hipMemcpyPeerAsync(...) (::sift_hip::LaunchGuard{}, (hipMemcpyPeerAsync)(__VA_ARGS__))
#define hipMemcpy(...) (::sift_hip::LaunchGuard{}, (hipMemcpy)(__VA_ARGS__))
#define hipMemsetAsync(...) (::sift_hip::LaunchGuard{}, (hipMemsetAsync)(__VA_ARGS__))
#define hipMemset(...) (::sift_hip::LaunchGuard{}, (hipMemset)(__VA_ARGS__))
#define hipEventRecord(...) (::sift_hip::LaunchGuard{}, (hipEventRecord)(__VA_ARGS__))
#define hipStreamWaitEvent(...) (::sift_hip::LaunchGuard{}, (hipStreamWaitEvent)(__VA_ARGS__))
namespace sift_hip {
// what a stage's launches left behind: an error the module launch path returned (launch_cache.h), else the runtime's own
hipError_t combined_last_error();
}
#define hipGetLastError() (::sift_hip::combined_last_error())
#ifdef __HIPCC__
#include <hip/hip_ext.h>
#include "launch_cache.h"
#undef hipLaunchKernelGGL
#ifdef SIFT_HIP_STATIC_LAUNCH   // the runtime's own launch path (a lookup of the host stub per launch): A/B builds only
#define hipLaunchKernelGGL(kernelName, ...)                                   \
    do {                                                                      \
        ::sift_hip::LaunchGuard sift_launch_guard_;                           \
        hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);                \
    } while (0)
#define hipExtLaunchKernelGGL(...)                                            \
    do {                                                                      \
        ::sift_hip::LaunchGuard sift_launch_guard_;                           \
        (hipExtLaunchKernelGGL)(__VA_ARGS__);                                 \
    } while (0)
#else
// every launch of the library: function object resolved once per device, module launch (launch_cache.h)
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...) \
    ::sift_hip::launch_cached((kernelName), dim3(numBlocks), dim3(numThreads), (unsigned)(memPerBlock), (streamId), nullptr, nullptr, ##__VA_ARGS__)
#define hipExtLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, startEvent, stopEvent, flags, ...) \
    ::sift_hip::launch_cached((kernelName), dim3(numBlocks), dim3(numThreads), (unsigned)(memPerBlock), (streamId), (startEvent), (stopEvent), ##__VA_ARGS__)
#endif
#endif

namespace sift_hip {

// Measurement switches inside kernels (options desc_dbg, orient_dbg, the cleanup stamps): phases switched off for timing - the
// results are then WRONG.  They exist in the ablation build only (-DSIFT_HIP_ABLATE, `make ablate`); in the shipped library and in
// libsift_hip_diag.so (whose kernels ARE the shipped ones) the mask is 0 and the compiler removes every branch that tests them.
#ifdef SIFT_HIP_ABLATE
constexpr int kDiagMask = ~0;
#else
constexpr int kDiagMask = 0;
#endif

constexpr int kMaxOctaves = 16;
constexpr int kMaxDogs = 16;                       // dogsPerEpoch upper bound of this build
constexpr int kMaxLevels = kMaxOctaves * (kMaxDogs + 1);
constexpr int kMaxRadiusFused = 32;                // blur radii with a fused LDS-tiled kernel
constexpr int kRegion = 8;                         // sift.cpp:61,164 `region`

// Extrema candidate as the scan emits it (order: octave, dog, x outer, y inner; sift.cpp:352-373)
struct Candidate {
    uint16_t x, y;
    uint16_t octave, index;
};
static_assert(sizeof(Candidate) == 8, "Candidate packing");

// Device-resident description of the current plan (one per context, uploaded when the plan
// changes).  Level buffers hold all n images of the batch back to back.
struct DevPlan {
    int n_images;
    int octaves, dogs;               // O, D  (levels per octave: D + 1 gaussians, D dogs)
    int w[kMaxOctaves], h[kMaxOctaves];
    float* gauss[kMaxLevels];        // [o * (D+1) + j]
    float* dog[kMaxLevels];          // [o * D + j]
    float* mag[kMaxLevels];          // gradient maps, null unless the level is selected
    float* ori[kMaxLevels];
    float* prod[kMaxLevels];         // magnitude * gaussian of the initial maps (orientationHistogram36 weight)
    float* w16[kMaxLevels];          // n x 256: top-left 16x16 of convolveWithGauss(level, 1.6)
    float gauss_scale[kMaxLevels];
    float dog_scale[kMaxLevels];
    // _findNearestGaussian(dog_scale[o*D+i]) for every (octave, dog): level id o'*(D+1)+i'
    int nearest_level[kMaxLevels];
    // radius of the dead 16x16 blur (sift.cpp:184) per (octave, dog); > 15 => the reference throws
    int dead_blur_radius[kMaxLevels];
    // extrema bitmask geometry: per scanned (octave, middle dog) "scan level" s
    int n_scan;
    int scan_octave[kMaxLevels], scan_dog[kMaxLevels];
    int scan_nyb[kMaxLevels];        // 64-row blocks per column
    int scan_word_base[kMaxLevels];  // first mask word of the level inside one image
    int words_per_image;
    // the fused scan's tiles (32 columns x one 64-row block) numbered in CANDIDATE order at strip granularity - scan level,
    // 32-column strip, 64-row block: a strip's candidates follow those of all tiles numbered before its first one
    int scan_tiles_x[kMaxLevels];    // 32-column strips of the level
    int scan_strip_base[kMaxLevels]; // first strip of the level among the image's strips
    int scan_tile_base[kMaxLevels];  // first tile of the level among the image's tiles
    int strips_per_image, tiles_per_image;
    long long cand_capacity;         // per image
    // grid of 16 px cells over the levels some keypoint scale selects (descriptor kernels): cells across / down, first cell of the level
    int desc_cell_base[kMaxLevels], desc_cw[kMaxLevels], desc_ch[kMaxLevels];
    int desc_cells_per_image;
};

#define SIFT_HIP_CHECK(expr)                                                        \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) throw ::sift_hip::HipError(_e, #expr, __FILE__, __LINE__); \
    } while (0)

struct HipError {
    hipError_t code;
    const char* expr;
    const char* file;
    int line;
    HipError(hipError_t c, const char* e, const char* f, int l) : code(c), expr(e), file(f), line(l) {}
};

// ---- records exchanged between host glue and kernels ------------------------------------------
struct OrientOut {           // per keypoint entering _orientationAssignment
    float orientation;       // *peaks.begin()
    uint16_t npeaks;         // peaks.size()
    uint8_t filtered;        // border test (sift.cpp:173-178)
    uint8_t throws;          // dead blur would throw (sift.cpp:184): 0 no, 1 kernel longer, 2 sigma < 0
};
static_assert(sizeof(OrientOut) == 8, "OrientOut packing");

// One keypoint of the orientation stage, in processing (spatial) order
struct OrientIn {
    uint16_t x, y;
    uint16_t octave, index;
    uint32_t kp;             // position in the survivor list (where the result goes)
    uint32_t pad;
};
static_assert(sizeof(OrientIn) == 16, "OrientIn packing");

struct FinalKp {             // one entry per keypoint entering _createDecriptors, vector order
    uint32_t cand;
    float orientation;
    uint16_t x, y;
    uint16_t octave, index;
};
static_assert(sizeof(FinalKp) == 16, "FinalKp packing");

// ---- kernel launchers (defined in the .hip files) ------------------------------------------------
// min_waves: smallest launch, in waves, that takes the streaming form (kStreamMinWaves unless the context's option
// "stream_min_waves" says otherwise: a value of the calling context, not of the process)
constexpr int kStreamMinWaves = 1024;  // below one wave per SIMD the tile kernel's finer work units win
void launch_blur(hipStream_t s, bool fused, const float* in, float* tmp, float* out, float* dog, int w,
                 int h, int n, const float* d_taps, int radius, int min_waves = kStreamMinWaves, hipEvent_t ev_start = nullptr,
                 hipEvent_t ev_stop = nullptr);
bool launch_blur_reduce(hipStream_t s, const float* in, float* dst, int w, int h, int wd, int hd, int n, const float* d_taps,
                        int radius, const int* d_inv_x, const int* d_inv_y, float* d_dump, int min_waves = kStreamMinWaves,
                        hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void set_stream_waves(int v);      // measurement build: < 0 restores the default; 0 = tile kernel only (process-wide)
void set_orient_dbg(int v);        // timing ablations of the orientation kernel (results are wrong when != 0)
void launch_resample(hipStream_t s, const float* src, float* dst, int ws, int hs, int wd, int hd, int n,
                     const int* d_lutx, const int* d_luty);
void launch_dog(hipStream_t s, const float* lower, const float* higher, float* out, size_t count);
bool launch_blur_reduce_kept(hipStream_t s, const float* in, float* dst, int w, int h, int wd, int hd, int n, const float* d_taps, int radius,
                             int sx, int sy, int min_waves, hipEvent_t ev_start, hipEvent_t ev_stop);
void launch_widen_u8(hipStream_t s, const uint8_t* in, float* out, size_t count);
void launch_io_copy(hipStream_t s, const void* src, void* dst, size_t bytes);      // kernels_io.hip: device-to-device copies as small kernels (group.cpp)
void launch_zero_ints(hipStream_t s, int* p, size_t n);
// one empty launch per translation unit: makes the runtime build that unit's device code (see kernels_*.hip, sift_hip_create)
void tu_touch_pyramid(hipStream_t s);
void tu_touch_pair(hipStream_t s);
// g0 = blur(in) and g1 = blur(g0) in one launch (kernels_pair.hip); false: not a shape that kernel takes
bool launch_blur_pair(hipStream_t s, const float* in, float* g0, float* g1, int w, int h, int n, const float* taps,
                      int radius, int min_waves, int pair_waves, hipEvent_t ev_start, hipEvent_t ev_stop);
void tu_touch_reduce(hipStream_t s);
void tu_touch_extrema(hipStream_t s);
void tu_touch_orient(hipStream_t s);
void tu_touch_desc(hipStream_t s);
void tu_touch_cleanup(hipStream_t s);
void tu_touch_wire(hipStream_t s);
void tu_touch_io(hipStream_t s);
   // instead of hipMemsetAsync between kernels of a stream
void launch_extrema_mask(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan,
                         unsigned long long* d_masks, int* d_counts);
void launch_extrema_scan(hipStream_t s, const DevPlan& plan, int* d_counts, int* d_totals);
bool extrema_edge_supported(const DevPlan& plan);
// every scan level of the plan in one launch (from_gauss: the scan forms its DoG tiles from four Gaussian levels)
void launch_extrema_edge(hipStream_t s, const DevPlan& plan, unsigned long long* d_masks, unsigned long long* d_fmasks,
                         int* d_tile_counts, bool from_gauss = false, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// candidate records, flag bytes and per-image totals from the fused scan's words and per-tile counts: no scan launch in between
void launch_extrema_expand_tiles(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan, const unsigned long long* d_masks,
                                 const unsigned long long* d_fmasks, const int* d_tile_counts, Candidate* d_cands, uint8_t* d_flags,
                                 int* d_totals);
constexpr int kFxCols = 32;   // columns of a tile of the fused scan (kernels_extrema.hip)
int resident_cus();   // CUs of the calling thread's device (kernels_pyramid.hip; 256 until sift_hip_create has asked)
void launch_extrema_expand(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan,
                           const unsigned long long* d_masks, const int* d_offsets, Candidate* d_cands,
                           const unsigned long long* d_fmasks = nullptr, uint8_t* d_flags = nullptr);
void launch_edge_filter(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan, const Candidate* d_cands,
                        const int* d_totals, uint8_t* d_flags);
void launch_edge_filter_points(hipStream_t s, const float* d0, const float* d1, const float* d2, int w, int h,
                               const uint16_t* xs, const uint16_t* ys, int m, uint8_t* flags);
void launch_vertex_parabola(hipStream_t s, const uint16_t* lnx, const float* lny, const uint16_t* px,
                            const float* py, const uint16_t* rnx, const float* rny, int m, float* out);
void launch_gradient(hipStream_t s, const float* g, float* mag, float* ori, float* prod, int w, int h,
                     int n, int* d_any_bin = nullptr, int stamp = 1, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_orientation(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan, const Candidate* d_cands,
                        const OrientIn* d_oin, const int* d_list_cnt, int list_cap, OrientOut* d_out,
                        float* d_peaks, int* d_next_group, const int* d_any_bin = nullptr, int zero_counters = 1, int stamp = 1);
// host-glue path: orientation inputs in list order from an uploaded survivor list
void launch_build_orient_in(hipStream_t s, const Candidate* d_cands, long long cand_cap, const uint32_t* d_list,
                            const int* d_list_cnt, int list_cap, int n_images, OrientIn* d_oin);
void launch_cleanup1(hipStream_t s, int n_images, const uint8_t* d_flags, const int* d_totals, long long cand_cap,
                     uint8_t* wk, uint32_t* wi, uint32_t* wi2, uint32_t* wp, uint32_t* d_list, OrientIn* d_oin,
                     uint32_t* d_lrank, const Candidate* d_cands, int list_cap, int* d_list_cnt, int* d_late_cnt,
                     int* d_fallback);
void launch_orient_prepare(hipStream_t s, int n_images, const uint8_t* d_flags, const int* d_totals, long long cand_cap,
                           int* d_chunk_cnt, const Candidate* d_cands, int list_cap, OrientIn* d_oin, int* d_early_cnt);
size_t orient_prepare_chunks(long long cand_cap);
void launch_cleanup2(hipStream_t s, int n_images, const Candidate* d_cands, long long cand_cap,
                     const uint32_t* d_list, const int* d_list_cnt, int list_cap, const OrientOut* d_orient,
                     const uint32_t* d_lrank, uint8_t* wk, uint32_t* wi, uint32_t* wi2, uint32_t* wp, FinalKp* d_final,
                     int* d_final_cnt, int* d_status, FinalKp* d_recs);
void cleanup_set_stamp_buffer(unsigned long long* d);
void launch_cleanup_kat(hipStream_t s, const uint8_t* d_flags, int n, uint8_t* wk, uint32_t* wi, uint32_t* wi2,
                        uint32_t* wp, uint32_t* d_out, int* d_info, int force_global, OrientIn* d_ord,
                        uint32_t* d_lrank, const Candidate* d_cd);
void launch_w16(hipStream_t s, const DevPlan& plan, int level, const float* d_taps16, int radius16);
void launch_desc_grid(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan, const FinalKp* d_final, const int* d_final_cnt,
                      int final_cap, int* d_cell_cnt, int* d_cell_off, FinalKp* d_pool, int pool_cap, long long* d_out_base,
                      bool compute_base, sift_hip_keypoint* d_kp_out, float* d_desc_out, long long out_cap);
void launch_descriptors_wave(hipStream_t s, const DevPlan* d_plan, const DevPlan& plan, int level, const int* d_cell_off,
                             const FinalKp* d_pool, int pool_cap, const long long* d_out_base, sift_hip_keypoint* d_kp_out,
                             float* d_desc_out, long long out_cap, int dbg = 0, int* d_wire_sums = nullptr, hipEvent_t ev_start = nullptr,
                             hipEvent_t ev_stop = nullptr);

// wire format of the keypoint gather (kernels_wire.hip)
size_t wire_blocks(long long total);
void launch_wire_unpack(hipStream_t s, const uint8_t* d_records, const float* d_values, long long total, int* d_sums,
                        long long* d_block_off, sift_hip_keypoint* d_kp, float* d_desc);
void launch_wire_count(hipStream_t s, const float* d_desc, long long total, int* d_sums, long long* d_block_off, bool counted);
void launch_wire_emit(hipStream_t s, const sift_hip_keypoint* d_kp, const float* d_desc, long long total,
                      const long long* d_block_off, uint8_t* d_records, float* d_values);

}  // namespace sift_hip
