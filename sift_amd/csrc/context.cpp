// libsift_hip.so host side: context, plan (pyramid geometry, scale schedule, tap tables), stage
// orchestration of Sift::calculate() (/root/reference/sift.cpp:19-57) and the C ABI of
// include/sift_hip.h.  All per-pixel and per-keypoint arithmetic runs in the HIP kernels; the
// host keeps only the reference's order-defining glue (std::sort cleanup, u16 truncation,
// appending extra orientation peaks).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <immintrin.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "phase_gate.h"
#include "host_glue.h"

using namespace sift_hip;

namespace {

constexpr int kPoolCap = 65536;  // records of the descriptor stage's cell grid: every keypoint once (u16_t size, sift.cpp:53)
constexpr int kListCap = 65536;  // cleanup keeps at most 65535 points (u16_t size, sift.cpp:41)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        ApiGuard api;
        if (p) SIFT_HIP_CHECK(hipFree(p));
        p = nullptr;
        cap = 0;
        SIFT_HIP_CHECK(hipMalloc(&p, bytes));
        cap = bytes;
    }
    void release() {
        ApiGuard api;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct HostBuf {  // pinned
    void* p = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        ApiGuard api;
        if (p) SIFT_HIP_CHECK(hipHostFree(p));
        p = nullptr;
        cap = 0;
        SIFT_HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        cap = bytes;
    }
    void release() {
        ApiGuard api;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct BlurOp {       // one alg::convolveWithGauss of the pyramid
    int kind;         // 0 subpixel pre-blur, 1 g(0,0), 2 level blur (+DoG), 3 reduce blur
    int octave, j;    // destination level for kind 2; source octave for kind 3
    float sigma;
    int radius;
    size_t tap_off;   // float offset into the tap table
    int w, h;         // image size the blur runs on
};

struct PointRec {     // host mirror of one vector<InterestPoint> element between stages
    uint32_t cand;
    float orientation;
    uint8_t filtered;
};

struct Plan {
    bool valid = false;
    int n = 0, in_w = 0, in_h = 0;
    sift_hip_params params{};
    int bw = 0, bh = 0;  // base image (after optional 2x upsampling)
    int O = 0, D = 0;
    DevPlan dev{};
    std::vector<BlurOp> ops;
    std::vector<float> taps;          // all tap tables back to back
    std::vector<int> luts;            // all index maps back to back
    std::vector<size_t> lut_x_off, lut_y_off;  // per octave transition (index o -> o+1); [O] = subpixel
    std::vector<size_t> inv_x_off, inv_y_off;  // inverse maps (source -> destination or -1) of the decimations
    std::vector<int> red_sx, red_sy;           // decimation o -> o+1: first destination column / row kept from source index 2 i + 1 (-1: the map has another form)
    bool dogs_carved = true;          // the arena holds the DoG levels (false: the fused extremum scan forms its DoG tiles from Gaussian levels)
    int fail_status = 0;              // plan-time precondition failure (depends only on sizes/params)
    size_t fail_op = (size_t)-1;      // first op that cannot run
    std::string fail_msg;
    std::vector<int> grad_levels;     // levels some keypoint scale selects
    std::vector<float> taps16;        // taps of convolveWithGauss(level, 1.6f) (sift.cpp:87)
    int radius16 = 0;
    size_t max_level_floats = 0;      // per image, largest level
};

}  // namespace

struct sift_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;          // side stream: work that can overlap the 1-block-per-image cleanup
    hipEvent_t ev_fork0 = nullptr, ev_fork = nullptr, ev_join = nullptr, ev_grad = nullptr;
    sift_hip::PhaseGate* gate = nullptr;   // shared with the other contexts of a BatchPipeline, or null
    struct sift_hip_gate* gate_owner = nullptr;
    long long gate_ticket = -1;             // this batch's ticket while it runs
    bool fused = true;
    bool fused_edge = true;   // extremum scan and edge filter in one LDS-tiled pass
    bool fused_reduce = true; // reduceToNextLevel: blur and decimation in one pass
    bool reduce_kept = true;  // ... that evaluates the kept pixels only (option "reduce_kept"; 0: blur_stream_kernel<..., DEC>)
    bool orient_general = false;  // tests: orientation histogram with per-sample bins even when every bin is 0
    bool gpu_cleanup = true;
    bool profile = false;     // this batch's blur launches carry timing events
    int profile_every = 0;    // option "profile": 0 off, N > 0: every N-th batch is timed (the events cost ~10 us per launch)
    long long profile_batches = 0;
    bool described = false;   // ... and their descriptors are already computed
    long long out_cap = 0;    // keypoints d_kp / d_desc hold
    int host_threads = 0;
    int desc_dbg = 0;
    Plan plan;
    DevBuf arena, d_plan, d_taps, d_luts, d_taps16, d_input, d_input_u8, d_base, d_tmp, d_tmp2;
    DevBuf d_sparse_rec, d_sparse_val;   // sift_hip_result_copy_sparse: the packed lists on their way to the host
    DevBuf d_masks, d_fmasks, d_counts, d_tile_counts, d_totals, d_cands, d_flags;
    DevBuf d_wk, d_wi, d_wi2, d_wp, d_status, d_pool;
    DevBuf d_order;
    HostBuf h_stage[2];              // pinned staging of pageable caller memory (host <-> device in chunks)
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    DevBuf d_cell_cnt, d_cell_off;   // descriptor grid: keypoints per 16 px cell, exclusive scan (+ total)
    int diag_repeat = 1;             // option "diag_repeat" (diagnostics, sift_hip_calculate_batch_device only)
    int gate_schedule = 1;           // option "gate_schedule" (phase_gate.h; 1 since round 3): applies to the gate this context is joined to
    // option "pyramid_side" (default on): the top Gaussian level of an octave (it only feeds the octave's last DoG) is formed on
    // the side stream, beside the reduction and the first levels of the next octave, which are too small to fill the chip alone
    bool pyramid_side = true;
    // option "dog_in_extrema" (round 5): the pyramid writes Gaussian levels only; the fused extremum scan fetches four of them per
    // scan level and forms its three DoG tiles on the way into LDS (128.0f + (g[j+1] - g[j]): alg::dog's two roundings); a DoG level
    // a caller asks for (sift_hip_level_copy) or the unfused scan needs is formed then
    bool dog_in_extrema = true;
    // option "blur_pair" (round 5, default on): the first two levels of the pyramid, g(0,0) and g(0,1), in one launch (kernels_pair.hip:
    // the second blur reads the first one's rows from LDS) - when no DoG level is written (dog_in_extrema), the two levels share
    // their taps and the batch fills the chip; otherwise two launches as before
    bool blur_pair = true;
    bool dogs_missing = false;       // this batch's DoG levels have not been written (the plan then has no room for them) ...
    int dog_in_scratch = -1;         // ... except the one a caller asked for last (sift_hip_level_copy forms it in d_tmp2)
    int stream_min_waves = 0;        // option "stream_min_waves" (0: the default, 1024): smallest launch, in waves, that takes the streaming blur
    int pair_waves = 0;              // option "pair_waves" (0: the default, 1536): waves the launch of the first two levels is cut into
    hipEvent_t ev_side_fork = nullptr, ev_side_join = nullptr;
    int bin_stamp = 0;               // this batch's value of the gradient pass's "some bin != 0" flags (kernels_orient.hip: launch_gradient)
    DevBuf d_list, d_list_cnt, d_orient, d_peaks, d_final, d_final_cnt, d_out_base, d_kp, d_desc;
    DevBuf d_wire_sums, d_wire_off;   // sparse wire format: floats per block of keypoints, their exclusive scan
    DevBuf d_unpack_sums, d_unpack_off;   // the same for sift_hip_sparse_unpack (lists that arrive from other GPUs)
    HostBuf h_wire;
    long long wire_values = -1, wire_for_total = -1;
    bool wire_count = false;          // option "wire_count": the descriptor kernel also counts the floats of the sparse wire format
    bool wire_counted = false;        // ... and has done so for the current batch
    bool wire_scanned = false;        // ... and the block offsets + the number of floats were queued behind it (h_wire holds them once the batch is done)
    int warm_calls = 0;               // calculate calls this context has completed (the first ones run alone: FirstBatch)
    int warm_copies = 0;              // ... and calls that fetch results (their first transfers)
    hipEvent_t ev_pack = nullptr;     // sift_hip_result_sparse_pack_async: recorded behind the pack kernel on the side stream
    bool pack_pending = false;        // ... and not yet waited for by this context's main stream
    DevBuf d_lrank, d_ochunk, d_ocnt, d_recs;   // list position -> orientation result; kept counts per 1024 candidates; early/late counts
    HostBuf h_flags, h_orient, h_peaks, h_status;
    hipEvent_t ev_sync = nullptr;
    bool spin_wait = true;    // poll an event instead of sleeping in hipStreamSynchronize (tens of microseconds per batch)
    // diagnostics (options "diag_pyramid_span", "diag_serial_gradient", "diag_cleanup_stamps"): all off by default
    bool diag_pyramid_span = false, diag_serial_gradient = false, diag_cleanup_stamps = false;
    // results of the last batch
    std::vector<int32_t> status, counts;
    std::vector<std::string> messages;
    std::vector<int> totals;                            // candidates per image
    std::vector<size_t> flag_off;                       // offset of image's flags in h_flags
    std::vector<std::vector<uint32_t>> list1;           // after first cleanup
    std::vector<std::vector<PointRec>> after_orient;    // after _orientationAssignment
    std::vector<std::vector<PointRec>> final_list;      // after second cleanup
    std::vector<long long> out_base;
    long long total = 0;
    long long last_total = 0;   // of the previous batch: sizes the output arrays before this batch's counts are known
    bool have_result = false;
    bool have_pyramid = false;
    bool stages_on_host = false;
    // profiling
    struct EvPair { hipEvent_t a, b; int which; double bytes; };
    std::vector<EvPair> pending;
    std::vector<hipEvent_t> event_pool;
    // classes: 0 the fused blur launches (streaming / tile / kept-pixels reduction / the pair launch), 1 the two-pass fallback,
    // 2 descriptor_wave_kernel, 3 extrema_edge_kernel, 4 the gradient maps' kernel (round 6: the E || D phase's three kernels)
    static constexpr int kProfClasses = 5;
    double prof_ms[kProfClasses] = {0, 0, 0, 0, 0};
    long long prof_launches[kProfClasses] = {0, 0, 0, 0, 0};
    double prof_bytes[kProfClasses] = {0, 0, 0, 0, 0};
    double prof_busy_ms[kProfClasses] = {0, 0, 0, 0, 0};   // time during which at least one launch of the class was running (union of the launches' intervals)
    long long prof_batches = 0;           // batches whose launches carried the events since the last reset
};

namespace {

void set_err(char* err, int errlen, const std::string& msg) {
    if (err && errlen > 0) {
        std::snprintf(err, (size_t)errlen, "%s", msg.c_str());
    }
}

std::string precondition(const char* text) { return std::string("Precondition violation!\n") + text; }

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

hipEvent_t get_event(sift_hip_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    ApiGuard api;
    SIFT_HIP_CHECK(hipEventCreate(&e));
    return e;
}

void resolve_events(sift_hip_ctx* c) {
    if (c->pending.size() >= 2 && c->diag_pyramid_span) {   // diagnostics: wall time of the blur chain vs the sum of its kernels
        float span = 0, sum = 0;
        (void)hipEventSynchronize(c->pending.back().b);
        (void)hipEventElapsedTime(&span, c->pending.front().a, c->pending.back().b);
        for (auto& p : c->pending) { float ms = 0; (void)hipEventElapsedTime(&ms, p.a, p.b); sum += ms; }
        std::fprintf(stderr, "pyramid span %.3f ms, sum of blur kernels %.3f ms, launches %zu\n", span, sum, c->pending.size());
    }
    std::vector<std::pair<float, float>> iv[sift_hip_ctx::kProfClasses];   // launch intervals, milliseconds after the batch's first launch began
    for (auto& p : c->pending) {
        float ms = 0, t0 = 0;
        SIFT_HIP_CHECK(hipEventSynchronize(p.b));
        SIFT_HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
        c->prof_ms[p.which] += ms;
        c->prof_launches[p.which] += 1;
        c->prof_bytes[p.which] += p.bytes;
        if (&p != &c->pending.front()) SIFT_HIP_CHECK(hipEventElapsedTime(&t0, c->pending.front().a, p.a));
        iv[p.which].emplace_back(t0, t0 + ms);
    }
    for (int wch = 0; wch < sift_hip_ctx::kProfClasses; ++wch) {   // launches on two streams overlap (option "pyramid_side"): count that time once
        std::sort(iv[wch].begin(), iv[wch].end());
        float end = -1e30f;
        for (auto& x : iv[wch]) {
            if (x.second <= end) continue;
            c->prof_busy_ms[wch] += x.second - std::max(x.first, end);
            end = x.second;
        }
    }
    for (auto& p : c->pending) {
        c->event_pool.push_back(p.a);
        c->event_pool.push_back(p.b);
    }
    c->pending.clear();
}

int min_waves(const sift_hip_ctx* c) { return c->stream_min_waves > 0 ? c->stream_min_waves : kStreamMinWaves; }

// Blur with optional event bracket.  Algorithmic bytes: 4 B read + 4 B written per
// pixel, + 4 B when the DoG is written too.
void run_blur(sift_hip_ctx* c, const float* in, float* out, float* dog, int w, int h, int n, size_t tap_off,
              int radius, hipStream_t st = nullptr) {
    const bool is_fused = c->fused && radius >= 1 && radius <= kMaxRadiusFused;
    hipEvent_t a = nullptr, b = nullptr;
    if (c->profile) {
        a = get_event(c);
        b = get_event(c);
    }
    // the events ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL): start/stop are the
    // kernel's begin/end timestamps and back-to-back launches stay back-to-back
    launch_blur(st ? st : c->stream, c->fused, in, c->d_tmp.as<float>(), out, dog, w, h, n, c->d_taps.as<float>() + tap_off,
                radius, min_waves(c), a, b);
    if (c->profile) {
        const double px = (double)w * (double)h * (double)n;
        c->pending.push_back({a, b, is_fused ? 0 : 1, px * (4.0 + (out ? 4.0 : 0.0) + (dog ? 4.0 : 0.0))});   // algorithmic bytes: read + levels written
    }
}

// ---- plan -------------------------------------------------------------------------------------------
// Mirrors Sift::_createDOGs' scale schedule (sift.cpp:388-411) and every Vigra precondition the
// pyramid can trip, in execution order.
int build_plan(sift_hip_ctx* c, int n, int w, int h, const sift_hip_params& prm, std::string& msg) {
    Plan& P = c->plan;
    // Does the batch write DoG levels?  Not when the fused extremum scan can take its DoG tiles from four Gaussian levels
    // (kernels_extrema.hip: rows of every scanned octave 16-byte aligned) - the plan then carves none (1 GB for 32 x 1080p).
    bool carve_dogs = !(c->dog_in_extrema && c->fused && c->fused_edge);
    {
        int ow = prm.subpixel ? 2 * w : w;
        for (int o = 0; o < prm.octaves && !carve_dogs; ++o, ow = (ow + 1) / 2)
            if (prm.dogs_per_epoch >= 3 && (ow % 4 != 0 || ow < 4)) carve_dogs = true;
    }
    if (P.valid && P.n == n && P.in_w == w && P.in_h == h && std::memcmp(&P.params, &prm, sizeof(prm)) == 0 && P.dogs_carved == carve_dogs)
        return SIFT_HIP_OK;
    P = Plan();
    P.dogs_carved = carve_dogs;
    if (n <= 0 || w <= 0 || h <= 0 || w >= 32768 || h >= 32768) {  // i16 loop counters, sift.cpp:354-355
        msg = "sift_hip: bad batch geometry";
        return SIFT_HIP_EINVAL;
    }
    if (!(prm.octaves > 0)) { msg = "Assertion `_octaves > 0' failed."; return SIFT_HIP_EASSERT; }
    if (!(prm.dogs_per_epoch >= 3)) { msg = "Assertion `_dogsPerEpoch >= 3' failed."; return SIFT_HIP_EASSERT; }
    if (prm.octaves > kMaxOctaves || prm.dogs_per_epoch > kMaxDogs) {
        msg = "sift_hip: this build supports at most 16 octaves and 16 DoGs per octave";
        return SIFT_HIP_EINVAL;
    }
    P.n = n; P.in_w = w; P.in_h = h; P.params = prm;
    const int O = P.O = prm.octaves, D = P.D = prm.dogs_per_epoch;
    DevPlan& dv = P.dev;
    std::memset(&dv, 0, sizeof(dv));
    dv.n_images = n; dv.octaves = O; dv.dogs = D;

    auto add_taps = [&](float sigma, int& radius, size_t& off) -> int {
        std::vector<float> t;
        if (!gauss_taps(sigma, t, radius)) return 2;
        off = P.taps.size();
        P.taps.insert(P.taps.end(), t.begin(), t.end());
        return 0;
    };
    auto fail = [&](const std::string& m) {
        if (P.fail_status == 0) {
            P.fail_status = SIFT_HIP_EPRECONDITION;
            P.fail_op = P.ops.size();
            P.fail_msg = m;
        }
    };
    // returns false when the op cannot run (precondition)
    auto add_blur = [&](int kind, int o, int j, float sigma, int bw, int bh) -> bool {
        BlurOp op{kind, o, j, sigma, 0, 0, bw, bh};
        if (add_taps(sigma, op.radius, op.tap_off) != 0) {
            fail(precondition("Kernel1D::initGaussian(): Standard deviation must be >= 0."));
            return false;
        }
        if (!(bw >= op.radius + 1)) { fail(precondition("separableConvolveX(): kernel longer than line\n")); return false; }
        if (!(bh >= op.radius + 1)) { fail(precondition("separableConvolveY(): kernel longer than line\n")); return false; }
        P.ops.push_back(op);
        return true;
    };
    auto add_lut = [&](int ws, int hs, int wd, int hd, size_t& xo, size_t& yo) -> bool {
        if (!(ws > 1 && hs > 1)) { fail(precondition("resizeImageNoInterpolation(): Source image too small.\n")); return false; }
        if (!(wd > 1 && hd > 1)) { fail(precondition("resizeImageNoInterpolation(): Destination image too small.\n")); return false; }
        const std::vector<int> lx = resize_index_map(ws, wd), ly = resize_index_map(hs, hd);
        xo = P.luts.size();
        P.luts.insert(P.luts.end(), lx.begin(), lx.end());
        yo = P.luts.size();
        P.luts.insert(P.luts.end(), ly.begin(), ly.end());
        return true;
    };
    // source index -> destination index (or -1) of a strictly increasing map, for the decimating blur
    auto add_inverse = [&](size_t map_off, int nd, int ns, size_t& inv_off) {
        std::vector<int> inv((size_t)ns, -1);
        bool ok_inv = true;
        for (int i = 0; i < nd; ++i) {
            const int sidx = P.luts[map_off + (size_t)i];
            if (sidx < 0 || sidx >= ns || inv[(size_t)sidx] != -1) { ok_inv = false; break; }
            inv[(size_t)sidx] = i;
        }
        if (!ok_inv) { inv_off = (size_t)-1; return; }
        inv_off = P.luts.size();
        P.luts.insert(P.luts.end(), inv.begin(), inv.end());
    };

    P.lut_x_off.assign((size_t)O + 1, 0);
    P.lut_y_off.assign((size_t)O + 1, 0);
    P.inv_x_off.assign((size_t)O + 1, (size_t)-1);
    P.inv_y_off.assign((size_t)O + 1, (size_t)-1);
    P.red_sx.assign((size_t)O + 1, -1);
    P.red_sy.assign((size_t)O + 1, -1);
    // lut[i] == 2 i for i < split, 2 i + 1 from there on (the form resizeImageNoInterpolation's map takes for a halving): split, or -1
    auto split_point = [&](size_t map_off, int nd) -> int {
        int split = nd;
        for (int i = 0; i < nd; ++i) {
            const int par = P.luts[map_off + (size_t)i] - 2 * i;
            if (par == 1 && split == nd) split = i;
            if (par != (i >= split ? 1 : 0)) return -1;
        }
        return split;
    };
    P.bw = w; P.bh = h;
    bool ok = true;
    if (prm.subpixel) {  // sift.cpp:20-21: increaseToNextLevel(img, 1.0)
        ok = add_blur(0, 0, 0, 1.0f, w, h) && add_lut(w, h, 2 * w, 2 * h, P.lut_x_off[(size_t)O], P.lut_y_off[(size_t)O]);
        if (ok) { P.bw = 2 * w; P.bh = 2 * h; }
    }
    for (int o = 0; o < O; ++o) {
        dv.w[o] = o == 0 ? P.bw : (dv.w[o - 1] + 1) / 2;
        dv.h[o] = o == 0 ? P.bh : (dv.h[o - 1] + 1) / 2;
    }
    // scale schedule, sift.cpp:388-411 (u16 exp, pow in double, product with float sigma in double)
    {
        dv.gauss_scale[0] = prm.sigma;
        uint16_t exp = 0;
        for (int i = 0; i < O; ++i) {
            for (int j = 1; j < D + 1; ++j) {
                const float scale = (float)(std::pow((double)prm.k, (double)exp) * (double)prm.sigma);
                dv.gauss_scale[i * (D + 1) + j] = scale;
                dv.dog_scale[i * D + j - 1] = scale - dv.gauss_scale[i * (D + 1) + j - 1];
                exp++;
            }
            if (i < O - 1) {
                dv.gauss_scale[(i + 1) * (D + 1)] = dv.gauss_scale[i * (D + 1) + D - 1];
                exp -= 2;
            }
        }
    }
    if (ok) ok = add_blur(1, 0, 0, prm.sigma, dv.w[0], dv.h[0]);
    for (int o = 0; o < O && ok; ++o) {
        for (int j = 1; j < D + 1 && ok; ++j) ok = add_blur(2, o, j, dv.gauss_scale[o * (D + 1) + j], dv.w[o], dv.h[o]);
        if (ok && o < O - 1) {
            ok = add_blur(3, o, D - 1, dv.gauss_scale[o * (D + 1) + D - 1], dv.w[o], dv.h[o]) &&
                 add_lut(dv.w[o], dv.h[o], dv.w[o + 1], dv.h[o + 1], P.lut_x_off[(size_t)o], P.lut_y_off[(size_t)o]);
            if (ok) {
                add_inverse(P.lut_x_off[(size_t)o], dv.w[o + 1], dv.w[o], P.inv_x_off[(size_t)o]);
                add_inverse(P.lut_y_off[(size_t)o], dv.h[o + 1], dv.h[o], P.inv_y_off[(size_t)o]);
                P.red_sx[(size_t)o] = split_point(P.lut_x_off[(size_t)o], dv.w[o + 1]);
                P.red_sy[(size_t)o] = split_point(P.lut_y_off[(size_t)o], dv.h[o + 1]);
            }
        }
    }
    // _findNearestGaussian for every DoG scale (sift.cpp:205-218); dead 16x16 blur (sift.cpp:184)
    for (int o = 0; o < O; ++o)
        for (int i = 0; i < D; ++i) {
            const float scale = dv.dog_scale[o * D + i];
            float lowest = 100;
            int best = 0;
            for (int oo = 0; oo < O; ++oo)
                for (int ii = 0; ii < D + 1; ++ii) {
                    const float cur = std::abs(dv.gauss_scale[oo * (D + 1) + ii] - scale);
                    if (cur < lowest) { lowest = cur; best = oo * (D + 1) + ii; }
                }
            dv.nearest_level[o * D + i] = best;
            const float s = (float)(1.5 * (double)scale);
            std::vector<float> t;
            int r = 0;
            if (!gauss_taps(s, t, r)) dv.dead_blur_radius[o * D + i] = 2;
            else dv.dead_blur_radius[o * D + i] = (16 >= r + 1) ? 0 : 1;
        }
    // scanned DoG levels and mask geometry
    dv.n_scan = 0;
    int words = 0;
    long long cap = 0;
    for (int o = 0; o < O; ++o)
        for (int i = 1; i < D - 1; ++i) {
            const int s = dv.n_scan++;
            dv.scan_octave[s] = o; dv.scan_dog[s] = i;
            dv.scan_nyb[s] = (dv.h[o] + 63) / 64;
            dv.scan_word_base[s] = words;
            words += dv.w[o] * dv.scan_nyb[s];
            dv.scan_tiles_x[s] = (dv.w[o] + kFxCols - 1) / kFxCols;
            dv.scan_strip_base[s] = dv.strips_per_image;
            dv.scan_tile_base[s] = dv.tiles_per_image;
            dv.strips_per_image += dv.scan_tiles_x[s];
            dv.tiles_per_image += dv.scan_tiles_x[s] * dv.scan_nyb[s];
            cap += (long long)std::max(0, dv.w[o] - 2) * (long long)std::max(0, dv.h[o] - 2);
            if (std::find(P.grad_levels.begin(), P.grad_levels.end(), dv.nearest_level[o * D + i]) == P.grad_levels.end())
                P.grad_levels.push_back(dv.nearest_level[o * D + i]);
        }
    {
        int cb = 0;
        for (int lvl : P.grad_levels) {   // 16 px cells (descriptor kernels)
            const int o = lvl / (D + 1);
            dv.desc_cw[lvl] = (dv.w[o] + 15) / 16;
            dv.desc_ch[lvl] = (dv.h[o] + 15) / 16;
            dv.desc_cell_base[lvl] = cb;
            cb += dv.desc_cw[lvl] * dv.desc_ch[lvl];
        }
        dv.desc_cells_per_image = std::max(cb, 1);
    }
    dv.words_per_image = std::max(words, 1);
    dv.cand_capacity = (std::max(cap, 1LL) + 15) / 16 * 16;   // every image's flag bytes start 16-byte aligned
    gauss_taps(1.6f, P.taps16, P.radius16);

    // ---- device memory ----------------------------------------------------------------------------
    size_t total = 0;
    std::vector<size_t> goff((size_t)O * (D + 1)), doff((size_t)O * D), moff(P.grad_levels.size()), ooff(P.grad_levels.size()), woff(P.grad_levels.size()), poff(P.grad_levels.size());
    P.max_level_floats = (size_t)std::max(w * (size_t)h, (size_t)P.bw * (size_t)P.bh);
    auto carve = [&](size_t floats) { const size_t o = total; total = align_up(total + floats * sizeof(float), 256); return o; };
    for (int o = 0; o < O; ++o) {
        const size_t px = (size_t)dv.w[o] * (size_t)dv.h[o] * (size_t)n;
        for (int j = 0; j < D + 1; ++j) goff[(size_t)(o * (D + 1) + j)] = carve(px);
        if (P.dogs_carved)
            for (int j = 0; j < D; ++j) doff[(size_t)(o * D + j)] = carve(px);
    }
    for (size_t g = 0; g < P.grad_levels.size(); ++g) {
        const int o = P.grad_levels[g] / (D + 1);
        const size_t px = (size_t)dv.w[o] * (size_t)dv.h[o] * (size_t)n;
        moff[g] = carve(px);
        ooff[g] = carve(px);
        woff[g] = carve((size_t)256 * (size_t)n);
        poff[g] = carve(px);
    }
    c->arena.ensure(total + 256);   // slack: the kept-pixels decimating blur may read the float after a level's last row (kernels_reduce.hip)
    char* base = c->arena.as<char>();
    for (size_t l = 0; l < goff.size(); ++l) dv.gauss[l] = reinterpret_cast<float*>(base + goff[l]);
    for (size_t l = 0; l < doff.size(); ++l) dv.dog[l] = P.dogs_carved ? reinterpret_cast<float*>(base + doff[l]) : nullptr;
    if (!P.dogs_carved && !extrema_edge_supported(dv)) {   // (cannot happen: the widths were checked above and every carve is 256-byte aligned)
        msg = "sift_hip: plan without DoG levels for a shape the fused extremum scan does not take";
        return SIFT_HIP_EINVAL;
    }
    for (size_t g = 0; g < P.grad_levels.size(); ++g) {
        dv.mag[P.grad_levels[g]] = reinterpret_cast<float*>(base + moff[g]);
        dv.ori[P.grad_levels[g]] = reinterpret_cast<float*>(base + ooff[g]);
        dv.w16[P.grad_levels[g]] = reinterpret_cast<float*>(base + woff[g]);
        dv.prod[P.grad_levels[g]] = reinterpret_cast<float*>(base + poff[g]);
    }
    const size_t lvl_bytes = P.max_level_floats * (size_t)n * sizeof(float);
    c->d_tmp.ensure(lvl_bytes);
    c->d_tmp2.ensure(lvl_bytes);
    if (prm.subpixel) c->d_base.ensure((size_t)P.bw * (size_t)P.bh * (size_t)n * sizeof(float));
    c->d_plan.ensure(sizeof(DevPlan));
    c->d_taps.ensure(std::max<size_t>(P.taps.size(), 1) * sizeof(float));
    c->d_luts.ensure(std::max<size_t>(P.luts.size(), 1) * sizeof(int));
    c->d_taps16.ensure(P.taps16.size() * sizeof(float));
    SIFT_HIP_CHECK(hipMemcpyAsync(c->d_plan.p, &dv, sizeof(DevPlan), hipMemcpyHostToDevice, c->stream));
    if (!P.taps.empty())
        SIFT_HIP_CHECK(hipMemcpyAsync(c->d_taps.p, P.taps.data(), P.taps.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (!P.luts.empty())
        SIFT_HIP_CHECK(hipMemcpyAsync(c->d_luts.p, P.luts.data(), P.luts.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    SIFT_HIP_CHECK(hipMemcpyAsync(c->d_taps16.p, P.taps16.data(), P.taps16.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const size_t nw = (size_t)dv.words_per_image * (size_t)n;
    c->d_masks.ensure(nw * sizeof(unsigned long long));
    c->d_fmasks.ensure(nw * sizeof(unsigned long long));
    c->d_counts.ensure(nw * sizeof(int));
    c->d_tile_counts.ensure((size_t)std::max(dv.tiles_per_image, 1) * (size_t)n * sizeof(int));
    c->d_totals.ensure((size_t)n * sizeof(int));
    c->d_cands.ensure((size_t)dv.cand_capacity * (size_t)n * sizeof(Candidate));
    c->d_flags.ensure((size_t)dv.cand_capacity * (size_t)n);
    c->d_list.ensure((size_t)kListCap * (size_t)n * sizeof(uint32_t));
    c->d_list_cnt.ensure((size_t)n * sizeof(int));
    c->d_order.ensure((size_t)kListCap * (size_t)n * sizeof(OrientIn));
    c->d_lrank.ensure((size_t)kListCap * (size_t)n * sizeof(uint32_t));
    c->d_recs.ensure((size_t)kListCap * (size_t)n * sizeof(FinalKp));
    c->d_ochunk.ensure(orient_prepare_chunks(dv.cand_capacity) * (size_t)n * sizeof(int));
    c->d_ocnt.ensure((size_t)n * 5 * sizeof(int));   // early counts, late counts, group counters of the two launches, bin flags
    c->d_orient.ensure((size_t)kListCap * (size_t)n * sizeof(OrientOut));
    c->d_peaks.ensure((size_t)kListCap * (size_t)n * 36 * sizeof(float));
    c->d_final.ensure((size_t)kListCap * (size_t)n * sizeof(FinalKp));
    c->d_final_cnt.ensure((size_t)n * sizeof(int));
    c->d_out_base.ensure((size_t)n * sizeof(long long));
    c->d_pool.ensure((size_t)kPoolCap * (size_t)n * sizeof(FinalKp));
    c->d_cell_cnt.ensure((size_t)(dv.desc_cells_per_image + 1) * (size_t)n * sizeof(int));
    c->d_cell_off.ensure((size_t)(dv.desc_cells_per_image + 1) * (size_t)n * sizeof(int));
    SIFT_HIP_CHECK(hipStreamSynchronize(c->stream));
    P.valid = true;
    return SIFT_HIP_OK;
}

// The top-left 16x16 of convolveWithGauss(level, 1.6) (sift.cpp:87) only needs the level itself: one small workgroup per
// image on the side stream as soon as the level exists, long before the descriptor stage asks for it.
void early_w16(sift_hip_ctx* c, int level) {
    Plan& P = c->plan;
    if (std::find(P.grad_levels.begin(), P.grad_levels.end(), level) == P.grad_levels.end()) return;
    SIFT_HIP_CHECK(hipEventRecord(c->ev_fork0, c->stream));
    SIFT_HIP_CHECK(hipStreamWaitEvent(c->stream2, c->ev_fork0, 0));
    launch_w16(c->stream2, P.dev, level, c->d_taps16.as<float>(), P.radius16);
}

// ---- pyramid (Sift::_createDOGs, sift.cpp:381-417) ---------------------------------------------
void run_pyramid(sift_hip_ctx* c, const float* d_in) {
    Plan& P = c->plan;
    const DevPlan& dv = P.dev;
    const int n = P.n, O = P.O, D = P.D;
    const float* base = d_in;
    bool side_used = false;
    hipStream_t ms = c->stream;   // stream of the launches that are not the side stream's
    struct SideJoin {   // whatever ran on the side stream is part of the pyramid: the main stream goes on after it
        sift_hip_ctx* c; bool* used;
        ~SideJoin() {
            if (!*used) return;
            (void)hipEventRecord(c->ev_side_join, c->stream2);
            (void)hipStreamWaitEvent(c->stream, c->ev_side_join, 0);
        }
    } side_join{c, &side_used};
    for (size_t k = 0; k < P.ops.size(); ++k) {
        if (k >= P.fail_op) break;
        const BlurOp& op = P.ops[k];
        switch (op.kind) {
            case 0: {  // increaseToNextLevel(img, 1.0): blur then 2x nearest upsample
                run_blur(c, d_in, c->d_tmp2.as<float>(), nullptr, op.w, op.h, n, op.tap_off, op.radius);
                launch_resample(c->stream, c->d_tmp2.as<float>(), c->d_base.as<float>(), op.w, op.h, P.bw, P.bh, n,
                                c->d_luts.as<int>() + P.lut_x_off[(size_t)O], c->d_luts.as<int>() + P.lut_y_off[(size_t)O]);
                base = c->d_base.as<float>();
                break;
            }
            case 1: {
                // option "blur_pair": g(0,0) and g(0,1) in one launch (kernels_pair.hip) when no DoG level is written; the
                // second blur then reads g(0,0) from LDS instead of HBM
                if (c->blur_pair && c->fused && c->dogs_missing && k + 1 < P.ops.size() && k + 1 < P.fail_op) {
                    const BlurOp& nx = P.ops[k + 1];
                    if (nx.kind == 2 && nx.octave == 0 && nx.j == 1 && nx.radius == op.radius && nx.w == op.w && nx.h == op.h &&
                        std::memcmp(P.taps.data() + op.tap_off, P.taps.data() + nx.tap_off, sizeof(float) * (size_t)(2 * op.radius + 1)) == 0) {
                        hipEvent_t a = nullptr, b = nullptr;
                        if (c->profile) { a = get_event(c); b = get_event(c); }
                        if (launch_blur_pair(c->stream, base, dv.gauss[0], dv.gauss[1], op.w, op.h, n, c->d_taps.as<float>() + op.tap_off,
                                             op.radius, min_waves(c), c->pair_waves, a, b)) {
                            if (c->profile) c->pending.push_back({a, b, 0, (double)op.w * (double)op.h * (double)n * 12.0});   // one read, two levels written
                            early_w16(c, 0);
                            early_w16(c, 1);
                            ++k;   // the next op was this launch's second half
                            break;
                        }
                        if (c->profile) { c->event_pool.push_back(a); c->event_pool.push_back(b); }
                    }
                }
                run_blur(c, base, dv.gauss[0], nullptr, op.w, op.h, n, op.tap_off, op.radius);
                early_w16(c, 0);
                break;
            }
            case 2: {
                const int l = op.octave * (D + 1) + op.j;
                // The top level of an octave only feeds the octave's last DoG; the next octave starts from the level below it
                // (sift.cpp:406-409).  Option "pyramid_side": it is formed on the side stream beside the reduction and the
                // next octave's levels (fused kernels only: the two-pass fallback shares a scratch image with them).
                const bool side = c->pyramid_side && c->ev_side_fork && op.j == D && op.octave + 1 < O && c->fused && op.radius >= 1 &&
                                  op.radius <= kMaxRadiusFused && std::find(P.grad_levels.begin(), P.grad_levels.end(), l) == P.grad_levels.end();
                const bool no_dogs = c->dogs_missing;   // option dog_in_extrema: every Gaussian level is kept, no DoG level is written
                float* g_out = dv.gauss[l];
                float* dog_out = no_dogs ? nullptr : dv.dog[op.octave * D + op.j - 1];
                if (side) {
                    SIFT_HIP_CHECK(hipEventRecord(c->ev_side_fork, c->stream));
                    SIFT_HIP_CHECK(hipStreamWaitEvent(c->stream2, c->ev_side_fork, 0));
                    run_blur(c, dv.gauss[l - 1], g_out, dog_out, op.w, op.h, n, op.tap_off, op.radius, c->stream2);
                    side_used = true;
                } else {
                    run_blur(c, dv.gauss[l - 1], g_out, dog_out, op.w, op.h, n, op.tap_off, op.radius, ms);
                    early_w16(c, l);
                }
                break;
            }
            case 3: {  // reduceToNextLevel(g(o, D-1), g(o, D-1).scale)
                const int o = op.octave;
                const float* src = dv.gauss[o * (D + 1) + D - 1];
                float* dst = dv.gauss[(o + 1) * (D + 1)];
                bool done = false;
                if (c->fused && c->fused_reduce && P.inv_x_off[(size_t)o] != (size_t)-1 && P.inv_y_off[(size_t)o] != (size_t)-1) {
                    // blur and decimation in one pass: the full-resolution blurred image is never written
                    hipEvent_t a = nullptr, b = nullptr;
                    if (c->profile) { a = get_event(c); b = get_event(c); }
                    // kept pixels only (kernels_reduce.hip) where the index maps and the shape allow it, else the streaming blur
                    // that stores the kept quarter of a full-resolution result
                    if (c->reduce_kept && P.red_sx[(size_t)o] >= 0 && P.red_sy[(size_t)o] >= 0)
                        done = launch_blur_reduce_kept(ms, src, dst, op.w, op.h, dv.w[o + 1], dv.h[o + 1], n, c->d_taps.as<float>() + op.tap_off,
                                                       op.radius, P.red_sx[(size_t)o], P.red_sy[(size_t)o], std::min(min_waves(c), 256), a, b);
                    if (!done)
                        done = launch_blur_reduce(ms, src, dst, op.w, op.h, dv.w[o + 1], dv.h[o + 1], n,
                                              c->d_taps.as<float>() + op.tap_off, op.radius, c->d_luts.as<int>() + P.inv_x_off[(size_t)o],
                                              c->d_luts.as<int>() + P.inv_y_off[(size_t)o], c->d_tmp.as<float>(), min_waves(c), a, b);
                    if (c->profile) {
                        if (done) {
                            const double px = (double)op.w * (double)op.h * (double)n, pd = (double)dv.w[o + 1] * (double)dv.h[o + 1] * (double)n;
                            c->pending.push_back({a, b, 0, px * 4.0 + pd * 4.0});   // algorithmic bytes: source read + kept pixels written
                        } else {
                            c->event_pool.push_back(a);
                            c->event_pool.push_back(b);
                        }
                    }
                }
                if (!done) {
                    run_blur(c, src, c->d_tmp2.as<float>(), nullptr, op.w, op.h, n, op.tap_off, op.radius, ms);
                    launch_resample(ms, c->d_tmp2.as<float>(), dst, dv.w[o], dv.h[o], dv.w[o + 1],
                                    dv.h[o + 1], n, c->d_luts.as<int>() + P.lut_x_off[(size_t)o], c->d_luts.as<int>() + P.lut_y_off[(size_t)o]);
                }
                break;
            }
        }
    }
}

struct Cleanup1Arg { sift_hip_ctx* c; };
void cleanup1_fn(int img, void* a) {
    sift_hip_ctx* c = static_cast<Cleanup1Arg*>(a)->c;
    cleanup_survivors(c->h_flags.as<uint8_t>() + c->flag_off[(size_t)img], c->totals[(size_t)img], c->list1[(size_t)img]);
}

struct Cleanup2Arg { sift_hip_ctx* c; };
void cleanup2_fn(int img, void* a) {
    sift_hip_ctx* c = static_cast<Cleanup2Arg*>(a)->c;
    const Plan& P = c->plan;
    const std::vector<uint32_t>& l1 = c->list1[(size_t)img];
    std::vector<PointRec>& ao = c->after_orient[(size_t)img];
    std::vector<PointRec>& fin = c->final_list[(size_t)img];
    ao.clear();
    fin.clear();
    const OrientOut* oo = c->h_orient.as<OrientOut>() + (size_t)img * kListCap;
    const float* pk = c->h_peaks.as<float>() ? c->h_peaks.as<float>() + (size_t)img * kListCap * 36 : nullptr;
    // sift.cpp:168-178, 184: a point past the border test whose dead blur cannot run throws
    for (size_t k = 0; k < l1.size(); ++k) {
        if (!oo[k].filtered && oo[k].throws) {
            c->status[(size_t)img] = SIFT_HIP_EPRECONDITION;
            c->messages[(size_t)img] = oo[k].throws == 2
                ? precondition("Kernel1D::initGaussian(): Standard deviation must be >= 0.")
                : precondition("separableConvolveX(): kernel longer than line\n");
            return;
        }
    }
    ao.reserve(l1.size());
    std::vector<PointRec> additional;
    for (size_t k = 0; k < l1.size(); ++k) {
        PointRec p{l1[k], oo[k].orientation, oo[k].filtered};
        ao.push_back(p);
        if (!oo[k].filtered && oo[k].npeaks > 1 && pk) {
            // sift.cpp:194-199: `peaks.begin()++` yields begin(), so every peak is appended
            for (int j = 0; j < oo[k].npeaks; ++j) additional.push_back(PointRec{l1[k], pk[k * 36 + (size_t)j], 0});
        }
    }
    ao.insert(ao.end(), additional.begin(), additional.end());
    std::vector<uint8_t> flags(ao.size());
    for (size_t k = 0; k < ao.size(); ++k) flags[k] = ao[k].filtered;
    std::vector<uint32_t> surv;
    cleanup_survivors(flags.data(), (int)flags.size(), surv);  // sift.cpp:49-54
    fin.reserve(surv.size());
    for (uint32_t s : surv) fin.push_back(ao[s]);
    (void)P;
}

// ---- stages between the edge filter and the descriptors -----------------------------------------
// Host-glue path: flags down, std::sort on the host (host_glue.cpp), lists up.  Exact by
// construction (it IS libstdc++'s std::sort); used when option "gpu_cleanup" is 0 and as the
// fallback for the cases the GPU cleanup flags (introsort depth limit, several orientation peaks).
void mid_host(sift_hip_ctx* c) {
    Plan& P = c->plan;
    const DevPlan& dv = P.dev;
    const int n = P.n;
    hipStream_t s = c->stream;
    const DevPlan* dpl = c->d_plan.as<DevPlan>();
    SIFT_HIP_CHECK(hipMemcpyAsync(c->totals.data(), c->d_totals.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    SIFT_HIP_CHECK(hipStreamSynchronize(s));
    c->flag_off.assign((size_t)n, 0);
    size_t fl_total = 0;
    for (int i = 0; i < n; ++i) { c->flag_off[(size_t)i] = fl_total; fl_total += (size_t)c->totals[(size_t)i]; }
    c->h_flags.ensure(std::max<size_t>(fl_total, 1));
    for (int i = 0; i < n; ++i)
        if (c->totals[(size_t)i])
            SIFT_HIP_CHECK(hipMemcpyAsync(c->h_flags.as<uint8_t>() + c->flag_off[(size_t)i],
                                          c->d_flags.as<uint8_t>() + (size_t)i * (size_t)dv.cand_capacity,
                                          (size_t)c->totals[(size_t)i], hipMemcpyDeviceToHost, s));
    SIFT_HIP_CHECK(hipStreamSynchronize(s));
    const int threads = c->host_threads > 0 ? c->host_threads : (int)std::max(1u, std::thread::hardware_concurrency());
    Cleanup1Arg a1{c};
    parallel_for(n, threads, cleanup1_fn, &a1);

    std::vector<int> cnt1((size_t)n);
    for (int i = 0; i < n; ++i) {
        cnt1[(size_t)i] = (int)c->list1[(size_t)i].size();
        if (cnt1[(size_t)i])
            SIFT_HIP_CHECK(hipMemcpyAsync(c->d_list.as<uint32_t>() + (size_t)i * kListCap, c->list1[(size_t)i].data(),
                                          (size_t)cnt1[(size_t)i] * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    }
    SIFT_HIP_CHECK(hipMemcpyAsync(c->d_list_cnt.p, cnt1.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    SIFT_HIP_CHECK(hipStreamWaitEvent(s, c->ev_join, 0));
    launch_build_orient_in(s, c->d_cands.as<Candidate>(), dv.cand_capacity, c->d_list.as<uint32_t>(), c->d_list_cnt.as<int>(),
                           kListCap, n, c->d_order.as<OrientIn>());
    launch_orientation(s, dpl, dv, c->d_cands.as<Candidate>(), c->d_order.as<OrientIn>(), c->d_list_cnt.as<int>(), kListCap,
                       c->d_orient.as<OrientOut>(), c->d_peaks.as<float>(), c->d_ocnt.as<int>() + 3 * n, (c->orient_general ? nullptr : c->d_ocnt.as<int>() + 4 * n),
                       1, c->bin_stamp);
    c->h_orient.ensure((size_t)n * kListCap * sizeof(OrientOut));
    for (int i = 0; i < n; ++i)
        if (cnt1[(size_t)i])
            SIFT_HIP_CHECK(hipMemcpyAsync(c->h_orient.as<OrientOut>() + (size_t)i * kListCap,
                                          c->d_orient.as<OrientOut>() + (size_t)i * kListCap,
                                          (size_t)cnt1[(size_t)i] * sizeof(OrientOut), hipMemcpyDeviceToHost, s));
    SIFT_HIP_CHECK(hipStreamSynchronize(s));
    bool any_multi = false;
    for (int i = 0; i < n && !any_multi; ++i) {
        const OrientOut* oo = c->h_orient.as<OrientOut>() + (size_t)i * kListCap;
        for (int k = 0; k < cnt1[(size_t)i]; ++k)
            if (!oo[k].filtered && oo[k].npeaks > 1) { any_multi = true; break; }
    }
    if (any_multi) {
        c->h_peaks.ensure((size_t)n * kListCap * 36 * sizeof(float));
        for (int i = 0; i < n; ++i)
            if (cnt1[(size_t)i])
                SIFT_HIP_CHECK(hipMemcpyAsync(c->h_peaks.as<float>() + (size_t)i * kListCap * 36,
                                              c->d_peaks.as<float>() + (size_t)i * kListCap * 36,
                                              (size_t)cnt1[(size_t)i] * 36 * sizeof(float), hipMemcpyDeviceToHost, s));
        SIFT_HIP_CHECK(hipStreamSynchronize(s));
    } else {
        c->h_peaks.release();
    }
    Cleanup2Arg a2{c};
    parallel_for(n, threads, cleanup2_fn, &a2);

    std::vector<int> cnt2((size_t)n);
    std::vector<FinalKp> up;
    std::vector<Candidate> hc;
    for (int i = 0; i < n; ++i) {
        const auto& fin = c->final_list[(size_t)i];
        cnt2[(size_t)i] = c->status[(size_t)i] ? 0 : (int)fin.size();
        c->counts[(size_t)i] = cnt2[(size_t)i];
        if (!cnt2[(size_t)i]) continue;
        hc.resize((size_t)c->totals[(size_t)i]);
        SIFT_HIP_CHECK(hipMemcpyAsync(hc.data(), c->d_cands.as<Candidate>() + (size_t)i * (size_t)dv.cand_capacity,
                                      hc.size() * sizeof(Candidate), hipMemcpyDeviceToHost, s));
        SIFT_HIP_CHECK(hipStreamSynchronize(s));
        up.resize(fin.size());
        for (size_t k = 0; k < fin.size(); ++k) {
            const Candidate& cd = hc[fin[k].cand];
            up[k] = FinalKp{fin[k].cand, fin[k].orientation, cd.x, cd.y, cd.octave, cd.index};
        }
        SIFT_HIP_CHECK(hipMemcpyAsync(c->d_final.as<FinalKp>() + (size_t)i * kListCap, up.data(), up.size() * sizeof(FinalKp),
                                      hipMemcpyHostToDevice, s));
        SIFT_HIP_CHECK(hipStreamSynchronize(s));
    }
    SIFT_HIP_CHECK(hipMemcpyAsync(c->d_final_cnt.p, cnt2.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    c->stages_on_host = true;
}

void ensure_outputs(sift_hip_ctx* c, long long keypoints) {
    if (keypoints <= c->out_cap) return;
    if (c->pack_pending) SIFT_HIP_CHECK(hipEventSynchronize(c->ev_pack));   // the arrays about to be freed are still being read
    c->out_cap = 0;   // until both arrays exist at the new size (an allocation that throws leaves them freed)
    c->wire_counted = c->wire_scanned = false;   // counts of a stage that wrote the old arrays
    c->d_kp.ensure((size_t)keypoints * sizeof(sift_hip_keypoint));
    c->d_desc.ensure((size_t)keypoints * 128 * sizeof(float));
    c->out_cap = keypoints;
}

void launch_descriptor_stage(sift_hip_ctx* c, bool base_on_device = false) {
    Plan& P = c->plan;
    const DevPlan& dv = P.dev;
    if (c->pack_pending) {   // the previous batch's lists are still being packed from the arrays this stage rewrites
        SIFT_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_pack, 0));
        c->pack_pending = false;
    }
    // grid of 16 px cells over the final keypoints, then one wave per keypoint - or per tile of 2 x 2 cells (kernels_desc.hip)
    launch_desc_grid(c->stream, c->d_plan.as<DevPlan>(), dv, c->d_final.as<FinalKp>(), c->d_final_cnt.as<int>(), kListCap,
                     c->d_cell_cnt.as<int>(), c->d_cell_off.as<int>(), c->d_pool.as<FinalKp>(), kPoolCap, c->d_out_base.as<long long>(),
                     base_on_device, c->d_kp.as<sift_hip_keypoint>(), c->d_desc.as<float>(), c->out_cap);
    int* wire_sums = nullptr;
    c->wire_counted = false;
    if (c->wire_count) {   // multi-GPU jobs: the counting pass of the wire format rides in the descriptor kernel
        const size_t nb = wire_blocks(c->out_cap);
        c->d_wire_sums.ensure((nb + 2) * sizeof(int));
        launch_zero_ints(c->stream, c->d_wire_sums.as<int>(), nb + 2);
        wire_sums = c->d_wire_sums.as<int>();
        c->wire_counted = true;
    }
    for (int lvl : P.grad_levels) {
        hipEvent_t a = nullptr, b = nullptr;
        if (c->profile) { a = get_event(c); b = get_event(c); }
        launch_descriptors_wave(c->stream, c->d_plan.as<DevPlan>(), dv, lvl, c->d_cell_off.as<int>(), c->d_pool.as<FinalKp>(), kPoolCap,
                                c->d_out_base.as<long long>(), c->d_kp.as<sift_hip_keypoint>(), c->d_desc.as<float>(), c->out_cap, c->desc_dbg,
                                wire_sums, a, b);
        // (bytes: SURVEY 8(d)'s 3.5 KB per keypoint - the keypoint count is not on the host yet, the caller prices the launches)
        if (c->profile) c->pending.push_back({a, b, 2, 0.0});
    }
    c->wire_scanned = false;
    if (c->wire_counted) {
        // ... and the scan of the per-block counts, over the blocks of the arrays' CAPACITY (the number of keypoints is not on
        // the host yet; blocks behind the last keypoint count zero), with the number of floats and the bin-7 flag sent to the
        // host behind it: when the batch is done sift_hip_result_sparse_size has its answer without a pass or a wait of its own
        const size_t nbc = wire_blocks(c->out_cap);
        c->d_wire_off.ensure((nbc + 1) * sizeof(long long));
        c->h_wire.ensure(2 * sizeof(long long));
        launch_wire_count(c->stream, c->d_desc.as<float>(), c->out_cap, c->d_wire_sums.as<int>(), c->d_wire_off.as<long long>(), true);
        SIFT_HIP_CHECK(hipMemcpyAsync(c->h_wire.p, c->d_wire_off.as<long long>() + nbc, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
        SIFT_HIP_CHECK(hipMemcpyAsync(c->h_wire.as<long long>() + 1, c->d_wire_sums.as<int>(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
        c->wire_scanned = true;
    }
}

// Wait for everything queued on `s`.  Polling an event returns within a microsecond or two of the GPU finishing;
// the runtime's blocking wait adds a wake-up latency that is a visible share of a 4.5 ms batch.
void wait_stream(sift_hip_ctx* c, hipStream_t s) {
    if (!c->spin_wait) {
        SIFT_HIP_CHECK(hipStreamSynchronize(s));
        return;
    }
    SIFT_HIP_CHECK(hipEventRecord(c->ev_sync, s));
    for (;;) {
        const hipError_t e = hipEventQuery(c->ev_sync);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) SIFT_HIP_CHECK(e);
        __builtin_ia32_pause();
    }
}

bool mid_gpu(sift_hip_ctx* c) {
    Plan& P = c->plan;
    const DevPlan& dv = P.dev;
    const int n = P.n;
    hipStream_t s = c->stream;
    const DevPlan* dpl = c->d_plan.as<DevPlan>();
    const size_t per = (size_t)std::max<long long>(dv.cand_capacity, kListCap);
    c->d_wk.ensure(per * (size_t)n);
    c->d_wi.ensure(per * (size_t)n * sizeof(uint32_t));
    c->d_wi2.ensure(per * (size_t)n * sizeof(uint32_t));
    c->d_wp.ensure(per * (size_t)n * sizeof(uint32_t));
    c->d_status.ensure((size_t)n * 5 * sizeof(int));
    int* d_fb1 = c->d_status.as<int>() + (size_t)n * 4;
    int* d_late = c->d_ocnt.as<int>() + n;
    launch_cleanup1(s, n, c->d_flags.as<uint8_t>(), c->d_totals.as<int>(), dv.cand_capacity, c->d_wk.as<uint8_t>(),
                    c->d_wi.as<uint32_t>(), c->d_wi2.as<uint32_t>(), c->d_wp.as<uint32_t>(), c->d_list.as<uint32_t>(),
                    c->d_order.as<OrientIn>(), c->d_lrank.as<uint32_t>(), c->d_cands.as<Candidate>(), kListCap,
                    c->d_list_cnt.as<int>(), d_late, d_fb1);
    // the side stream has meanwhile run the orientation stage for every image without u16 truncation
    SIFT_HIP_CHECK(hipStreamWaitEvent(s, c->ev_join, 0));
    // late launch: only the images whose survivor list was truncated (counts are 0 for the others)
    launch_orientation(s, dpl, dv, c->d_cands.as<Candidate>(), c->d_order.as<OrientIn>(), d_late, kListCap,
                       c->d_orient.as<OrientOut>(), c->d_peaks.as<float>(), c->d_ocnt.as<int>() + 3 * n, (c->orient_general ? nullptr : c->d_ocnt.as<int>() + 4 * n),
                       0 /* its group counters were cleared together with the early launch's */, c->bin_stamp);
    launch_cleanup2(s, n, c->d_cands.as<Candidate>(), dv.cand_capacity, c->d_list.as<uint32_t>(), c->d_list_cnt.as<int>(),
                    kListCap, c->d_orient.as<OrientOut>(), c->d_lrank.as<uint32_t>(), c->d_wk.as<uint8_t>(),
                    c->d_wi.as<uint32_t>(), c->d_wi2.as<uint32_t>(), c->d_wp.as<uint32_t>(), c->d_final.as<FinalKp>(),
                    c->d_final_cnt.as<int>(), c->d_status.as<int>(), c->d_recs.as<FinalKp>());
    // The descriptor stage does not wait for the counts to reach the host: output slots come from a device-side
    // scan and the output arrays keep a generous capacity; should a batch ever exceed it, the (idempotent: the
    // mutated maps only live in LDS) stage is simply run again after growing them.
    // (the images' first output slots - a scan of the final counts - are formed by the descriptor stage's grid kernel itself)
    // capacity guess: a quarter above what this context's previous batch returned; before the first one 24576 keypoints
    // per image (a 1080p frame returns ~20 k; 13 MB of results per image)
    ensure_outputs(c, std::max<long long>(c->out_cap, c->last_total > 0 ? c->last_total + c->last_total / 4 : (long long)n * 24576));
    if (c->gate) c->gate->before_descriptors(c->gate_ticket, s);
    launch_descriptor_stage(c, true);
    SIFT_HIP_CHECK(hipGetLastError());
    if (c->gate) c->gate->mark(c->gate_ticket, sift_hip::PhaseGate::kD, s);
    c->described = true;
    c->h_status.ensure((size_t)n * 5 * sizeof(int));
    const int* st = c->h_status.as<int>();   // pinned: the copy is a real asynchronous DMA
    SIFT_HIP_CHECK(hipMemcpyAsync(c->h_status.p, c->d_status.p, (size_t)n * 5 * sizeof(int), hipMemcpyDeviceToHost, s));
    resolve_events(c);   // the pyramid's timing events completed long ago: read them while the GPU is still busy
    wait_stream(c, s);
    if (c->diag_cleanup_stamps) {   // diagnostics: phases of the second cleanup (image 0)
        static unsigned long long* dst = nullptr;
        if (!dst) { (void)hipMalloc(&dst, 512 * sizeof(unsigned long long)); (void)hipMemset(dst, 0, 512 * sizeof(unsigned long long)); cleanup_set_stamp_buffer(dst); }
        else {
            unsigned long long hs[512];
            (void)hipMemcpy(hs, dst, sizeof(hs), hipMemcpyDeviceToHost);
            std::fprintf(stderr, "cleanup2 (us): init %.1f sort %.1f compact+emit %.1f binning %.1f\n", (hs[101] - hs[100]) / 100.0,
                         (hs[102] - hs[101]) / 100.0, (hs[103] - hs[102]) / 100.0, (hs[104] - hs[103]) / 100.0);
            std::fprintf(stderr, "cleanup1 (us): bits %.1f ranks %.1f | rounds %llu:", (hs[9] > hs[8] ? (hs[9] - hs[8]) / 100.0 : 0.0) * 0 + (hs[8] ? 0.0 : 0.0), (hs[9] - hs[8]) / 100.0, hs[7]);
            for (unsigned long long r = 0; r < hs[7] && r < 40; ++r)
                std::fprintf(stderr, " [p%llu n%llu %.1f]", hs[16 + 4 * r + 3] >> 60, (hs[16 + 4 * r + 3] >> 32) & 0xfffffff,
                             r + 1 < hs[7] ? (hs[16 + 4 * (r + 1)] - hs[16 + 4 * r]) / 100.0 : 0.0);
            std::fprintf(stderr, "\n");
        }
    }
    for (int i = 0; i < n; ++i)
        if (st[(size_t)i * 4 + 1] || st[(size_t)n * 4 + (size_t)i]) {
            c->described = false;
            c->wire_counted = c->wire_scanned = false;   // the speculative descriptor stage's counts are not this batch's
            return false;
        }
    for (int i = 0; i < n; ++i) {
        c->counts[(size_t)i] = st[(size_t)i * 4 + 0];
        if (st[(size_t)i * 4 + 2] != 0x7fffffff) {  // sift.cpp:184 would throw for this image
            c->status[(size_t)i] = SIFT_HIP_EPRECONDITION;
            c->messages[(size_t)i] = st[(size_t)i * 4 + 3] == 2
                ? precondition("Kernel1D::initGaussian(): Standard deviation must be >= 0.")
                : precondition("separableConvolveX(): kernel longer than line\n");
            c->counts[(size_t)i] = 0;
        }
    }
    c->stages_on_host = false;
    return true;
}

// Fill the host-side stage vectors from the device buffers (inspection API after the GPU path).
void ensure_host_stages(sift_hip_ctx* c) {
    if (c->stages_on_host) return;
    Plan& P = c->plan;
    const DevPlan& dv = P.dev;
    const int n = P.n;
    SIFT_HIP_CHECK(hipMemcpy(c->totals.data(), c->d_totals.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    c->flag_off.assign((size_t)n, 0);
    size_t fl_total = 0;
    for (int i = 0; i < n; ++i) { c->flag_off[(size_t)i] = fl_total; fl_total += (size_t)c->totals[(size_t)i]; }
    c->h_flags.ensure(std::max<size_t>(fl_total, 1));
    std::vector<int> cnt1((size_t)n), cnt2((size_t)n);
    SIFT_HIP_CHECK(hipMemcpy(cnt1.data(), c->d_list_cnt.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    SIFT_HIP_CHECK(hipMemcpy(cnt2.data(), c->d_final_cnt.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<OrientOut> oo((size_t)kListCap);
    std::vector<uint32_t> lr;
    std::vector<FinalKp> fk;
    for (int i = 0; i < n; ++i) {
        if (c->totals[(size_t)i])
            SIFT_HIP_CHECK(hipMemcpy(c->h_flags.as<uint8_t>() + c->flag_off[(size_t)i],
                                     c->d_flags.as<uint8_t>() + (size_t)i * (size_t)dv.cand_capacity,
                                     (size_t)c->totals[(size_t)i], hipMemcpyDeviceToHost));
        auto& l1 = c->list1[(size_t)i];
        l1.resize((size_t)cnt1[(size_t)i]);
        lr.resize(l1.size());
        if (!l1.empty()) {
            SIFT_HIP_CHECK(hipMemcpy(l1.data(), c->d_list.as<uint32_t>() + (size_t)i * kListCap, l1.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            SIFT_HIP_CHECK(hipMemcpy(lr.data(), c->d_lrank.as<uint32_t>() + (size_t)i * kListCap, lr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            SIFT_HIP_CHECK(hipMemcpy(oo.data(), c->d_orient.as<OrientOut>() + (size_t)i * kListCap, oo.size() * sizeof(OrientOut), hipMemcpyDeviceToHost));
        }
        auto& ao = c->after_orient[(size_t)i];
        auto& fin = c->final_list[(size_t)i];
        ao.clear();
        fin.clear();
        if (c->status[(size_t)i]) continue;  // the reference threw inside _orientationAssignment
        for (size_t k = 0; k < l1.size(); ++k) ao.push_back(PointRec{l1[k], oo[lr[k]].orientation, oo[lr[k]].filtered});
        fk.resize((size_t)cnt2[(size_t)i]);
        if (!fk.empty())
            SIFT_HIP_CHECK(hipMemcpy(fk.data(), c->d_final.as<FinalKp>() + (size_t)i * kListCap, fk.size() * sizeof(FinalKp), hipMemcpyDeviceToHost));
        for (const FinalKp& f : fk) fin.push_back(PointRec{f.cand, f.orientation, 0});
    }
    c->stages_on_host = true;
}

// ---- host <-> device transfers of the boundary ---------------------------------------------------------------
// The reference's caller hands calculate() an image in ordinary memory (main.cpp:52-57).  A copy engine reads pinned
// memory at the PCIe rate but pageable memory only through the runtime's own bounce buffer, at a fraction of it, so:
//   * memory the caller got from sift_hip_host_alloc (or registered itself) goes to the device in one asynchronous copy;
//   * anything else is moved in 16 MB chunks through two pinned staging buffers, the host-side copy of chunk i+1
//     (several threads) running while the copy engine moves chunk i.
constexpr size_t kStageChunk = 16u << 20;

bool is_pinned(const void* p) {
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();   // ordinary memory: "invalid value", not an error of ours
        return false;
    }
    return a.type == hipMemoryTypeHost || a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

struct CopyJob { char* dst; const char* src; size_t bytes; int parts; };
void copy_part(int i, void* arg) {
    const CopyJob* j = static_cast<const CopyJob*>(arg);
    const size_t per = (j->bytes + (size_t)j->parts - 1) / (size_t)j->parts;
    const size_t lo = std::min(j->bytes, per * (size_t)i), hi = std::min(j->bytes, lo + per);
    if (hi > lo) std::memcpy(j->dst + lo, j->src + lo, hi - lo);
}
void host_copy(sift_hip_ctx* c, void* dst, const void* src, size_t bytes) {
    const int threads = std::max(1, std::min(c->host_threads > 0 ? c->host_threads : 4, (int)(bytes >> 20) + 1));
    CopyJob j{static_cast<char*>(dst), static_cast<const char*>(src), bytes, threads};
    parallel_for(threads, threads, copy_part, &j);
}

void ensure_staging(sift_hip_ctx* c) {
    for (int i = 0; i < 2; ++i) {
        c->h_stage[i].ensure(kStageChunk);
        if (!c->ev_stage[i]) { ApiGuard api; SIFT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_stage[i], hipEventDisableTiming)); }
    }
}

// host -> device on stream s; returns once `host` may be reused (pageable) or at once (pinned: the caller keeps the
// buffer until the batch is done, which calculate only returns after)
// One chunk between the context's page-locked staging buffer and device memory.  (Measured in rounds 2 - 3 and removed in
// round 4: the same moves, and the transfers of page-locked caller memory, as kernels of this library reading / writing the
// mapped host memory in place - they disturb the bandwidth-bound kernels less than the runtime's blit kernels but read the link
// at ~28 GB/s and lose overall, 6.5 against 4.5 - 5.5 ms per host-to-host batch; they did not lower the rate of the runtime's
// crashes in multi-threaded hosts either.)
void stage_move(sift_hip_ctx*, void* dst, const void* src, size_t bytes, bool to_device, hipStream_t s) {
    SIFT_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, s));
}

void upload(sift_hip_ctx* c, void* dev, const void* host, size_t bytes, hipStream_t s) {
    if (bytes == 0) return;
    if (is_pinned(host)) {
        SIFT_HIP_CHECK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s));
        return;
    }
    ensure_staging(c);
    size_t off = 0;
    for (int k = 0; off < bytes; ++k, off += kStageChunk) {
        const int b = k & 1;
        const size_t n = std::min(kStageChunk, bytes - off);
        if (k >= 2) SIFT_HIP_CHECK(hipEventSynchronize(c->ev_stage[b]));   // the copy engine is done with this buffer
        host_copy(c, c->h_stage[b].p, static_cast<const char*>(host) + off, n);
        stage_move(c, static_cast<char*>(dev) + off, c->h_stage[b].p, n, true, s);
        SIFT_HIP_CHECK(hipEventRecord(c->ev_stage[b], s));
    }
}

// device -> caller memory (host or device), complete on return
void download(sift_hip_ctx* c, void* dst, const void* dev, size_t bytes, hipStream_t s) {
    if (bytes == 0) return;
    if (is_pinned(dst)) {
        SIFT_HIP_CHECK(hipMemcpyAsync(dst, dev, bytes, hipMemcpyDefault, s));
        wait_stream(c, s);
        return;
    }
    ensure_staging(c);
    size_t off = 0;
    const size_t chunks = (bytes + kStageChunk - 1) / kStageChunk;
    for (size_t k = 0; k <= chunks; ++k) {
        if (k < chunks) {
            const int b = (int)(k & 1);
            const size_t n = std::min(kStageChunk, bytes - k * kStageChunk);
            stage_move(c, c->h_stage[b].p, static_cast<const char*>(dev) + k * kStageChunk, n, false, s);
            SIFT_HIP_CHECK(hipEventRecord(c->ev_stage[b], s));
        }
        if (k >= 1) {   // chunk k-1 has landed (or lands now) while chunk k is on its way
            const int b = (int)((k - 1) & 1);
            const size_t n = std::min(kStageChunk, bytes - off);
            SIFT_HIP_CHECK(hipEventSynchronize(c->ev_stage[b]));
            host_copy(c, static_cast<char*>(dst) + off, c->h_stage[b].p, n);
            off += n;
        }
    }
}

int run_batch(sift_hip_ctx* c, const float* d_in, char* err, int errlen) {
    Plan& P = c->plan;
    const DevPlan& dv = P.dev;
    const int n = P.n;
    hipStream_t s = c->stream;
    c->status.assign((size_t)n, 0);
    c->counts.assign((size_t)n, 0);
    c->messages.assign((size_t)n, std::string());
    c->totals.assign((size_t)n, 0);
    c->list1.assign((size_t)n, {});
    c->after_orient.assign((size_t)n, {});
    c->final_list.assign((size_t)n, {});
    c->out_base.assign((size_t)n, 0);
    c->total = 0;
    c->have_result = false;
    c->have_pyramid = false;
    c->stages_on_host = false;
    c->described = false;
    c->wire_values = c->wire_for_total = -1;
    c->wire_counted = false;
    c->wire_scanned = false;
    c->profile = c->profile_every > 0 && (c->profile_batches++ % c->profile_every) == 0;
    if (c->profile) c->prof_batches++;

    // Batches of several contexts in flight on this GPU: the gate orders their phases (phase_gate.h).  Whatever
    // this batch owes its partners is released when the scope ends, however it ends.
    struct GateScope {
        sift_hip_ctx* c;
        ~GateScope() {
            if (c->gate && c->gate_ticket >= 0) c->gate->finish(c->gate_ticket, c->stream);
            c->gate_ticket = -1;
        }
    } gate_scope{c};
    c->gate_ticket = c->gate ? c->gate->begin_batch(s) : -1;
    c->dog_in_scratch = -1;
    c->dogs_missing = !P.dogs_carved;
    run_pyramid(c, d_in);
    SIFT_HIP_CHECK(hipGetLastError());   // a rejected launch configuration must not go unnoticed
    if (c->gate) {
        c->gate->mark(c->gate_ticket, sift_hip::PhaseGate::kP, s);
    }
    c->have_pyramid = true;
    if (P.fail_status) {
        SIFT_HIP_CHECK(hipStreamSynchronize(s));
        resolve_events(c);
        for (int i = 0; i < n; ++i) { c->status[(size_t)i] = P.fail_status; c->messages[(size_t)i] = P.fail_msg; }
        c->have_result = true;
        c->stages_on_host = true;
        set_err(err, errlen, P.fail_msg);
        return P.fail_status;
    }

    // Gradient maps and W16 only need the pyramid: side stream, from here on.  (Not earlier: sharing the
    // CUs with the HBM-bound blur kernels slows those by more than the overlap wins.)
    const DevPlan* dpl = c->d_plan.as<DevPlan>();
    const bool serial_gradient = c->diag_serial_gradient;   // diagnostics: no overlap
    hipStream_t gs = serial_gradient ? s : c->stream2;
    SIFT_HIP_CHECK(hipEventRecord(c->ev_fork0, s));
    SIFT_HIP_CHECK(hipStreamWaitEvent(c->stream2, c->ev_fork0, 0));
    // "some sample has a bin != 0" per image: flagged with this batch's stamp, never cleared (launch_gradient)
    c->bin_stamp = (c->bin_stamp % 0x3fffffff) + 1;
    for (int lvl : P.grad_levels) {
        const int o = lvl / (P.D + 1);
        hipEvent_t a = nullptr, b = nullptr;
        if (c->profile) { a = get_event(c); b = get_event(c); }
        launch_gradient(gs, dv.gauss[lvl], dv.mag[lvl], dv.ori[lvl], dv.prod[lvl], dv.w[o], dv.h[o], n,
                        (c->orient_general ? nullptr : c->d_ocnt.as<int>() + 4 * n), c->bin_stamp, a, b);
        if (c->profile) c->pending.push_back({a, b, 4, (double)dv.w[o] * (double)dv.h[o] * (double)n * 12.0});   // grad_mag_ori 12 N
    }
    if (c->gate) SIFT_HIP_CHECK(hipEventRecord(c->ev_grad, gs));
    // extrema + edge responses (sift.cpp:33-34)
    if (c->fused_edge && extrema_edge_supported(dv)) {
        // one pass over the DoG levels: extremum test and edge-response filter from LDS tiles
        hipEvent_t a = nullptr, b = nullptr;
        if (c->profile) { a = get_event(c); b = get_event(c); }
        launch_extrema_edge(s, dv, c->d_masks.as<unsigned long long>(), c->d_fmasks.as<unsigned long long>(), c->d_tile_counts.as<int>(), c->dogs_missing, a, b);
        if (c->profile) {   // 12 N per scanned middle level read as three DoG levels; 16 N when the scan reads four Gaussian levels
            double px = 0;
            for (int k = 0; k < dv.n_scan; ++k) px += (double)dv.w[dv.scan_octave[k]] * (double)dv.h[dv.scan_octave[k]];
            c->pending.push_back({a, b, 3, px * (double)n * (c->dogs_missing ? 16.0 : 12.0)});
        }
        // (no scan launch: the expansion finds a strip's first slot from the per-tile counts of the tiles in front of it)
        launch_extrema_expand_tiles(s, dpl, dv, c->d_masks.as<unsigned long long>(), c->d_fmasks.as<unsigned long long>(), c->d_tile_counts.as<int>(),
                                    c->d_cands.as<Candidate>(), c->d_flags.as<uint8_t>(), c->d_totals.as<int>());
    } else {
        launch_extrema_mask(s, dpl, dv, c->d_masks.as<unsigned long long>(), c->d_counts.as<int>());
        launch_extrema_scan(s, dv, c->d_counts.as<int>(), c->d_totals.as<int>());
        launch_extrema_expand(s, dpl, dv, c->d_masks.as<unsigned long long>(), c->d_counts.as<int>(), c->d_cands.as<Candidate>());
        launch_edge_filter(s, dpl, dv, c->d_cands.as<Candidate>(), c->d_totals.as<int>(), c->d_flags.as<uint8_t>());
    }
    // The orientation stage is order-independent: it follows the gradient maps on the side stream, over
    // the kept candidates in scan order, while the one-workgroup-per-image cleanup kernel emulates the
    // reference's sort on the main stream (sift.cpp:37-47).
    if (c->gate) {
        // end of the chip-filling stretch E (main stream and the gradient maps of the side stream), start of the
        // cleanup chain C
        if (gs != s) SIFT_HIP_CHECK(hipStreamWaitEvent(s, c->ev_grad, 0));
        c->gate->mark(c->gate_ticket, sift_hip::PhaseGate::kE, s);
        c->gate->before_cleanup(c->gate_ticket, s);
    }
    SIFT_HIP_CHECK(hipEventRecord(c->ev_fork, s));
    SIFT_HIP_CHECK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    if (c->gpu_cleanup) {
        launch_orient_prepare(c->stream2, n, c->d_flags.as<uint8_t>(), c->d_totals.as<int>(), dv.cand_capacity,
                              c->d_ochunk.as<int>(), c->d_cands.as<Candidate>(), kListCap, c->d_order.as<OrientIn>(),
                              c->d_ocnt.as<int>());
        launch_orientation(c->stream2, dpl, dv, c->d_cands.as<Candidate>(), c->d_order.as<OrientIn>(), c->d_ocnt.as<int>(),
                           kListCap, c->d_orient.as<OrientOut>(), c->d_peaks.as<float>(), c->d_ocnt.as<int>() + 2 * n, (c->orient_general ? nullptr : c->d_ocnt.as<int>() + 4 * n),
                           2 /* also the late launch's counters, which follow */, c->bin_stamp);
    }
    SIFT_HIP_CHECK(hipEventRecord(c->ev_join, c->stream2));
    SIFT_HIP_CHECK(hipGetLastError());
    // cleanup, orientation assignment, cleanup (sift.cpp:37-54)
    if (!(c->gpu_cleanup && mid_gpu(c))) {
        for (int i = 0; i < n; ++i) { c->status[(size_t)i] = 0; c->messages[(size_t)i].clear(); }
        mid_host(c);
    }
    resolve_events(c);

    // descriptors (sift.cpp:55)
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        c->out_base[(size_t)i] = total;
        total += c->counts[(size_t)i];
    }
    c->total = total;
    c->last_total = total;
    if (c->described && total > c->out_cap) c->described = false;   // outputs were too small: grow and redo
    if (!c->described) {
        ensure_outputs(c, std::max<long long>(total, 1));
        SIFT_HIP_CHECK(hipMemcpyAsync(c->d_out_base.p, c->out_base.data(), (size_t)n * sizeof(long long), hipMemcpyHostToDevice, s));
        if (total > 0) {
            launch_descriptor_stage(c);
        }
    }
    if (!c->described) wait_stream(c, s);   // (the GPU path has already waited for the whole batch)
    c->have_result = true;
    int rc = SIFT_HIP_OK;
    for (int i = 0; i < n; ++i)
        if (c->status[(size_t)i]) {
            rc = c->status[(size_t)i];
            set_err(err, errlen, c->messages[(size_t)i]);
            break;
        }
    return rc;
}

template <class F>
int guarded(char* err, int errlen, F&& f) {
    try {
        return f();
    } catch (const HipError& e) {
        char buf[512];
        std::snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d: %s", (int)e.code, hipGetErrorString(e.code), e.file, e.line, e.expr);
        set_err(err, errlen, buf);
        return SIFT_HIP_EHIP;
    } catch (const std::exception& e) {
        set_err(err, errlen, e.what());
        return SIFT_HIP_EHIP;
    }
}

}  // namespace

// =====================================================================================================
// Contexts that run batches side by side need hardware queues of their own: with the HIP runtime's default of 4, streams
// of different contexts share a queue and inherit each other's ordering (the phase gate then loses what it arranges).
// The runtime reads GPU_MAX_HW_QUEUES when it initialises, so this only has an effect if it comes before the process's
// first HIP call; a value the host has set is left alone.
// The library does not touch the environment (setenv is not safe beside a host's other threads): the HOST sets
// GPU_MAX_HW_QUEUES=8 before its first HIP call when it runs several contexts on one GPU (sift_amd/_lib.py does at import,
// the example programs do at the top of main).

extern "C" {

const char* sift_hip_version(void) { return "sift_hip 0.1 (gfx950)"; }

int sift_hip_create(int device, sift_hip_ctx** out, char* err, int errlen) {
    if (!out) return SIFT_HIP_EINVAL;
    *out = nullptr;
    return guarded(err, errlen, [&]() {
        int count = 0;
        SIFT_HIP_CHECK(hipGetDeviceCount(&count));
        if (device < 0 || device >= count) {
            set_err(err, errlen, "sift_hip_create: no such HIP device");
            return SIFT_HIP_EINVAL;
        }
        SIFT_HIP_CHECK(hipSetDevice(device));
        ApiGuard api;   // streams, events and the first launches: not beside another thread's launches (common.h)
        auto* c = new sift_hip_ctx();
        c->device = device;
        SIFT_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        SIFT_HIP_CHECK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        SIFT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fork0, hipEventDisableTiming));
        SIFT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        SIFT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        SIFT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_grad, hipEventDisableTiming));
        SIFT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_sync, hipEventDisableTiming));
        SIFT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_side_fork, hipEventDisableTiming));
        SIFT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_side_join, hipEventDisableTiming));
        {
            // Device code is built lazily, on the first launch from a translation unit, and that step does not survive two host
            // threads doing it at once: have it done here, once per device, before any worker thread can launch anything.
            static std::mutex touch_lock;
            static bool touched[64] = {false};
            std::lock_guard<std::mutex> lk(touch_lock);
            if (device < 64 && !touched[device]) {
                tu_touch_pyramid(c->stream); tu_touch_pair(c->stream); tu_touch_reduce(c->stream); tu_touch_extrema(c->stream); tu_touch_orient(c->stream);
                tu_touch_desc(c->stream); tu_touch_cleanup(c->stream); tu_touch_wire(c->stream); tu_touch_io(c->stream);
                SIFT_HIP_CHECK(hipGetLastError());
                SIFT_HIP_CHECK(hipStreamSynchronize(c->stream));
                touched[device] = true;
            }
        }
        *out = c;
        return SIFT_HIP_OK;
    });
}

void sift_hip_destroy(sift_hip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)sift_hip_set_gate(c, nullptr);
    ApiGuard api;
    for (DevBuf* b : {&c->arena, &c->d_plan, &c->d_taps, &c->d_luts, &c->d_taps16, &c->d_input, &c->d_input_u8, &c->d_sparse_rec, &c->d_sparse_val, &c->d_base, &c->d_tmp, &c->d_tmp2,
                      &c->d_wk, &c->d_wi, &c->d_wi2, &c->d_wp, &c->d_status, &c->d_pool, &c->d_order, &c->d_masks, &c->d_fmasks, &c->d_counts, &c->d_tile_counts, &c->d_totals, &c->d_cands, &c->d_flags, &c->d_list, &c->d_list_cnt, &c->d_orient,
                      &c->d_peaks, &c->d_final, &c->d_final_cnt, &c->d_out_base, &c->d_kp, &c->d_desc, &c->d_lrank, &c->d_ochunk, &c->d_ocnt, &c->d_recs, &c->d_wire_sums, &c->d_wire_off, &c->d_unpack_sums, &c->d_unpack_off, &c->d_cell_cnt, &c->d_cell_off})
        b->release();
    for (HostBuf* b : {&c->h_flags, &c->h_orient, &c->h_peaks, &c->h_status, &c->h_wire, &c->h_stage[0], &c->h_stage[1]}) b->release();
    for (auto& e : c->ev_stage) if (e) (void)hipEventDestroy(e);
    for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c->ev_fork0);
    (void)hipEventDestroy(c->ev_grad);
    (void)hipEventDestroy(c->ev_fork);
    (void)hipEventDestroy(c->ev_join);
    (void)hipEventDestroy(c->ev_sync);
    if (c->ev_pack) (void)hipEventDestroy(c->ev_pack);
    if (c->ev_side_fork) (void)hipEventDestroy(c->ev_side_fork);
    if (c->ev_side_join) (void)hipEventDestroy(c->ev_side_join);
    (void)hipStreamDestroy(c->stream2);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

struct sift_hip_gate {
    int device;
    sift_hip::PhaseGate gate;
    std::mutex m;
    int attached = 0;     // contexts joined by this gate (at most PhaseGate::kMaxContexts: its slot ring)
    bool released = false;   // sift_hip_gate_destroy has been called: the last context to leave frees the gate
};


int sift_hip_gate_create(int device, sift_hip_gate** out) {
    if (!out) return SIFT_HIP_EINVAL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return SIFT_HIP_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return SIFT_HIP_EHIP;
    auto* g = new sift_hip_gate();
    g->device = device;
    *out = g;
    return SIFT_HIP_OK;
}

void sift_hip_gate_destroy(sift_hip_gate* g) {
    if (!g) return;
    // contexts still joined by the gate keep it alive (they reach into it from run_batch and from sift_hip_destroy): the
    // gate goes when the last of them leaves, whichever order the host destroys things in
    {
        std::lock_guard<std::mutex> lk(g->m);
        g->released = true;
        if (g->attached > 0) return;
    }
    (void)hipSetDevice(g->device);
    delete g;
}

int sift_hip_set_gate(sift_hip_ctx* c, sift_hip_gate* g) {
    if (!c || (g && g->device != c->device)) return SIFT_HIP_EINVAL;
    if (c->gate_owner == g) return SIFT_HIP_OK;
    if (g) {
        std::lock_guard<std::mutex> lk(g->m);
        if (g->released) return SIFT_HIP_EINVAL;
        // more contexts than the gate's ring can tell apart would reuse the slot of a batch that is still live
        if (g->attached >= sift_hip::PhaseGate::kMaxContexts) return SIFT_HIP_EINVAL;
        g->attached++;
    }
    if (c->gate_owner) {
        bool last;
        {
            std::lock_guard<std::mutex> lk(c->gate_owner->m);
            last = --c->gate_owner->attached == 0 && c->gate_owner->released;
        }
        if (last) {
            (void)hipSetDevice(c->device);
            delete c->gate_owner;
        }
    }
    c->gate_owner = g;
    c->gate = g ? &g->gate : nullptr;
    if (c->gate) c->gate->set_schedule(c->gate_schedule);
    return SIFT_HIP_OK;
}

int sift_hip_set_option(sift_hip_ctx* c, const char* name, int value) {
    if (!c || !name) return SIFT_HIP_EINVAL;
    // ---- the eight options of the shipped library (include/sift_hip.h) ----
    if (!std::strcmp(name, "profile")) { c->profile_every = value > 0 ? value : 0; c->profile_batches = 0; c->profile = false; return SIFT_HIP_OK; }
    if (!std::strcmp(name, "wire_count")) { c->wire_count = value != 0; return SIFT_HIP_OK; }
    if (!std::strcmp(name, "host_threads")) { c->host_threads = value; return SIFT_HIP_OK; }
    if (!std::strcmp(name, "spin_wait")) { c->spin_wait = value != 0; return SIFT_HIP_OK; }
    if (!std::strcmp(name, "orient_general")) { c->orient_general = value != 0; return SIFT_HIP_OK; }
    if (!std::strcmp(name, "stream_min_waves")) { c->stream_min_waves = value > 0 ? value : 0; return SIFT_HIP_OK; }
    if (!std::strcmp(name, "blur_pair")) { c->blur_pair = value != 0; return SIFT_HIP_OK; }
    if (!std::strcmp(name, "pair_waves")) { c->pair_waves = value > 0 ? value : 0; return SIFT_HIP_OK; }
#ifdef SIFT_HIP_DIAG
    // ---- libsift_hip_diag.so (`make -C sift_amd/csrc diag`: this file alone compiled with -DSIFT_HIP_DIAG, the kernels are the
    // shipped ones).  Names that FORCE a fallback path the library otherwise takes by itself when a shape does not fit the
    // default kernels - tests/diag_fallbacks.py runs them against the oracle - and host-side measurement aids of tools/.
    if (!std::strcmp(name, "fused_blur")) { c->fused = value != 0; return SIFT_HIP_OK; }                 // 0: two-pass row / column kernels (radius 0 and > 32)
    if (!std::strcmp(name, "fused_edge")) { c->fused_edge = value != 0; return SIFT_HIP_OK; }            // 0: mask kernel + thread-per-candidate filter (rows not 16-byte aligned)
    if (!std::strcmp(name, "fused_reduce")) { c->fused_reduce = value != 0; return SIFT_HIP_OK; }        // 0: blur into a temporary, then the resampling kernel (index maps without an inverse)
    if (!std::strcmp(name, "reduce_kept")) { c->reduce_kept = value != 0; return SIFT_HIP_OK; }          // 0: the streaming blur that stores the kept quarter (maps without a parity split)
    if (!std::strcmp(name, "dog_in_extrema")) { c->dog_in_extrema = value != 0; return SIFT_HIP_OK; }    // 0: every blur launch writes its DoG level (rows not 16-byte aligned)
    if (!std::strcmp(name, "gpu_cleanup")) { c->gpu_cleanup = value != 0; return SIFT_HIP_OK; }          // 0: std::sort on the host (introsort depth limit, several orientation peaks)
    if (!std::strcmp(name, "pyramid_side")) { c->pyramid_side = value != 0; return SIFT_HIP_OK; }        // 0: every pyramid launch on one stream
    if (!std::strcmp(name, "gate_schedule")) {                                                           // 0: no pyramid shares the chip (rounds 1 - 2)
        if (value < 0 || value > 1) return SIFT_HIP_EINVAL;
        c->gate_schedule = value;
        if (c->gate) c->gate->set_schedule(value);
        return SIFT_HIP_OK;
    }
    if (!std::strcmp(name, "diag_repeat")) { c->diag_repeat = value > 1 ? value : 1; return SIFT_HIP_OK; }
    if (!std::strcmp(name, "stream_waves")) { set_stream_waves(value); return SIFT_HIP_OK; }
    if (!std::strcmp(name, "diag_pyramid_span")) { c->diag_pyramid_span = value != 0; return SIFT_HIP_OK; }
    if (!std::strcmp(name, "diag_serial_gradient")) { c->diag_serial_gradient = value != 0; return SIFT_HIP_OK; }
#endif
#ifdef SIFT_HIP_ABLATE
    // ---- libsift_hip_ablate.so (`make ablate`: every file with -DSIFT_HIP_DIAG -DSIFT_HIP_ABLATE): phases INSIDE kernels switched
    // off for timing - the results are then WRONG - and stamps.  Scripts under tools/ only.
    if (!std::strcmp(name, "desc_dbg")) { c->desc_dbg = value; return SIFT_HIP_OK; }
    if (!std::strcmp(name, "orient_dbg")) { set_orient_dbg(value); return SIFT_HIP_OK; }
    if (!std::strcmp(name, "diag_cleanup_stamps")) { c->diag_cleanup_stamps = value != 0; return SIFT_HIP_OK; }
#endif
    return SIFT_HIP_EINVAL;
}

}  // extern "C"

namespace {
// A context's FIRST batch runs alone in the process.  The HIP runtime sets a good deal up lazily, on first use - a kernel's
// function object at its first launch, a stream's copy machinery at its first transfer - and it is while several host threads
// go through those first uses side by side that the launches crash inside the runtime (SEGV below hipLaunchKernel: a launch
// that finds a null object; common.h, tools/example_loop.sh).  Every launch and every runtime copy of this library is under the
// device's launch lock anyway, which serialises the CALLS but not what they start; a whole first batch under one lock per
// DEVICE serialises that too (the first uses that collide are per device), and costs nothing once every context has run once.
std::mutex& first_batch_mutex(int device) {   // one per device: a first batch that waits device-wide - growing a buffer while this
    static std::mutex m[64];                  // shard's RCCL send is unmatched - must not hold up the other devices' shards, whose reports post the receive
    return m[(unsigned)device % 64u];
}
constexpr int kWarmCalls = 2;     // the second batch still meets first uses (e.g. buffers that only now grow, the gather's first copies)
struct FirstBatch {
    int* n;
    std::unique_lock<std::mutex> lk;
    FirstBatch(int* counter, int device) : n(counter) {
        if (*n < kWarmCalls) lk = std::unique_lock<std::mutex>(first_batch_mutex(device));
    }
    ~FirstBatch() { if (*n < kWarmCalls) ++*n; }
};
}  // namespace

extern "C" {

int sift_hip_calculate_batch_device(sift_hip_ctx* c, const void* dev_imgs, int n, int w, int h,
                                    const sift_hip_params* params, char* err, int errlen) {
    if (!c || !dev_imgs || !params) return SIFT_HIP_EINVAL;
    FirstBatch first_batch(&c->warm_calls, c->device);
    return guarded(err, errlen, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        c->have_result = c->have_pyramid = false;   // whatever happens next, the previous batch's results are gone
        std::string msg;
        const int rc = build_plan(c, n, w, h, *params, msg);
        if (rc) { set_err(err, errlen, msg); return rc; }
        // diagnostics (option "diag_repeat" = k > 1): the batch k times over without returning to the caller in between - what a
        // host with no turnaround between a context's batches would see
        for (int r = 1; r < c->diag_repeat; ++r) {
            const int rr = run_batch(c, static_cast<const float*>(dev_imgs), err, errlen);
            if (rr) return rr;
        }
        return run_batch(c, static_cast<const float*>(dev_imgs), err, errlen);
    });
}

int sift_hip_calculate_batch(sift_hip_ctx* c, const float* host_imgs, int n, int w, int h,
                             const sift_hip_params* params, char* err, int errlen) {
    if (!c || !host_imgs || !params || n <= 0 || w <= 0 || h <= 0) return SIFT_HIP_EINVAL;
    FirstBatch first_batch(&c->warm_calls, c->device);
    return guarded(err, errlen, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        c->have_result = c->have_pyramid = false;   // whatever happens next, the previous batch's results are gone
        const size_t bytes = (size_t)n * (size_t)w * (size_t)h * sizeof(float);
        c->d_input.ensure(bytes);
        upload(c, c->d_input.p, host_imgs, bytes, c->stream);
        std::string msg;
        const int rc = build_plan(c, n, w, h, *params, msg);
        if (rc) { set_err(err, errlen, msg); return rc; }
        return run_batch(c, c->d_input.as<float>(), err, errlen);
    });
}

// 8-bit frames: what a host that reads 8-bit files holds (the reference's inputs are such files, main.cpp:52-54).  A quarter
// of the bytes cross the link; the GPU widens them to the integer-valued floats vigra::importImage would have produced.
int sift_hip_calculate_batch_u8(sift_hip_ctx* c, const uint8_t* host_imgs, int n, int w, int h,
                                const sift_hip_params* params, char* err, int errlen) {
    if (!c || !host_imgs || !params || n <= 0 || w <= 0 || h <= 0) return SIFT_HIP_EINVAL;
    FirstBatch first_batch(&c->warm_calls, c->device);
    return guarded(err, errlen, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        c->have_result = c->have_pyramid = false;
        const size_t count = (size_t)n * (size_t)w * (size_t)h;
        c->d_input.ensure(count * sizeof(float));
        c->d_input_u8.ensure(count);
        upload(c, c->d_input_u8.p, host_imgs, count, c->stream);
        launch_widen_u8(c->stream, c->d_input_u8.as<uint8_t>(), c->d_input.as<float>(), count);
        SIFT_HIP_CHECK(hipGetLastError());
        std::string msg;
        const int rc = build_plan(c, n, w, h, *params, msg);
        if (rc) { set_err(err, errlen, msg); return rc; }
        return run_batch(c, c->d_input.as<float>(), err, errlen);
    });
}

int sift_hip_calculate_batch_device_u8(sift_hip_ctx* c, const void* dev_imgs, int n, int w, int h,
                                       const sift_hip_params* params, char* err, int errlen) {
    if (!c || !dev_imgs || !params || n <= 0 || w <= 0 || h <= 0) return SIFT_HIP_EINVAL;
    FirstBatch first_batch(&c->warm_calls, c->device);
    return guarded(err, errlen, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        c->have_result = c->have_pyramid = false;
        const size_t count = (size_t)n * (size_t)w * (size_t)h;
        c->d_input.ensure(count * sizeof(float));
        launch_widen_u8(c->stream, static_cast<const uint8_t*>(dev_imgs), c->d_input.as<float>(), count);
        SIFT_HIP_CHECK(hipGetLastError());
        std::string msg;
        const int rc = build_plan(c, n, w, h, *params, msg);
        if (rc) { set_err(err, errlen, msg); return rc; }
        return run_batch(c, c->d_input.as<float>(), err, errlen);
    });
}

void* sift_hip_host_alloc(size_t bytes) {
    void* p = nullptr;
    LaunchGuard api(current_device_refreshed());   // no context here: the lock of the device the HOST put this thread on
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void sift_hip_host_free(void* p) {
    LaunchGuard api(current_device_refreshed());
    if (p) (void)hipHostFree(p);
}

int sift_hip_result_images(sift_hip_ctx* c) { return (c && c->have_result) ? (int)c->status.size() : -1; }
int sift_hip_result_status(sift_hip_ctx* c, int32_t* status, int cap) {
    if (!c || !c->have_result || !status || cap < (int)c->status.size()) return SIFT_HIP_EINVAL;
    std::copy(c->status.begin(), c->status.end(), status);
    return SIFT_HIP_OK;
}
int sift_hip_result_counts(sift_hip_ctx* c, int32_t* counts, int cap) {
    if (!c || !c->have_result || !counts || cap < (int)c->counts.size()) return SIFT_HIP_EINVAL;
    std::copy(c->counts.begin(), c->counts.end(), counts);
    return SIFT_HIP_OK;
}
int64_t sift_hip_result_total(sift_hip_ctx* c) { return (c && c->have_result) ? c->total : -1; }

int sift_hip_result_copy(sift_hip_ctx* c, sift_hip_keypoint* kp, float* desc) {
    if (!c || !c->have_result) return SIFT_HIP_EINVAL;
    FirstBatch first_copy(&c->warm_copies, c->device);
    return guarded(nullptr, 0, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        if (c->total > 0) {
            if (kp) download(c, kp, c->d_kp.p, (size_t)c->total * sizeof(sift_hip_keypoint), c->stream);
            if (desc) download(c, desc, c->d_desc.p, (size_t)c->total * 128 * sizeof(float), c->stream);
        }
        return SIFT_HIP_OK;
    });
}
int sift_hip_result_device(sift_hip_ctx* c, const void** kp, const void** desc) {
    if (!c || !c->have_result) return SIFT_HIP_EINVAL;
    if (kp) *kp = c->d_kp.p;
    if (desc) *desc = c->d_desc.p;
    return SIFT_HIP_OK;
}

int sift_hip_result_sparse_size(sift_hip_ctx* c, int64_t* n_values, int* lossless) {
    if (!c || !c->have_result || !n_values) return SIFT_HIP_EINVAL;
    char err[256];
    if (c->total == 0) {   // nothing to send
        c->wire_values = 0;
        c->wire_for_total = 0;
        *n_values = 0;
        if (lossless) *lossless = 1;
        return SIFT_HIP_OK;
    }
    if (c->wire_scanned && c->described && c->total <= c->out_cap) {   // counted and scanned by the batch itself (option wire_count): the host already has the answer
        c->wire_values = *c->h_wire.as<long long>();
        if (lossless) *lossless = *reinterpret_cast<const int*>(c->h_wire.as<long long>() + 1) ? 0 : 1;
        c->wire_for_total = c->total;
        *n_values = c->wire_values;
        return SIFT_HIP_OK;
    }
    return guarded(err, sizeof(err), [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        const size_t nb = wire_blocks(c->total);
        const bool counted = c->wire_counted;   // by the last descriptor stage of this batch (launch_descriptor_stage)
        if (!counted) c->d_wire_sums.ensure((nb + 2) * sizeof(int));
        c->d_wire_off.ensure((nb + 1) * sizeof(long long));
        c->h_wire.ensure(2 * sizeof(long long));
        launch_wire_count(c->stream, c->d_desc.as<float>(), c->total, c->d_wire_sums.as<int>(), c->d_wire_off.as<long long>(), counted);
        SIFT_HIP_CHECK(hipMemcpyAsync(c->h_wire.p, c->d_wire_off.as<long long>() + nb, sizeof(long long), hipMemcpyDeviceToHost, c->stream));
        SIFT_HIP_CHECK(hipMemcpyAsync(c->h_wire.as<long long>() + 1, c->d_wire_sums.as<int>(), sizeof(int), hipMemcpyDeviceToHost, c->stream));
        wait_stream(c, c->stream);
        c->wire_values = *c->h_wire.as<long long>();
        if (lossless) *lossless = *reinterpret_cast<const int*>(c->h_wire.as<long long>() + 1) ? 0 : 1;
        c->wire_for_total = c->total;
        *n_values = c->wire_values;
        return SIFT_HIP_OK;
    });
}

int sift_hip_result_sparse_pack(sift_hip_ctx* c, void* d_records, void* d_values) {
    if (!c || !c->have_result || c->wire_for_total != c->total || c->wire_values < 0) return SIFT_HIP_EINVAL;
    if (c->total > 0 && (!d_records || (c->wire_values > 0 && !d_values))) return SIFT_HIP_EINVAL;
    char err[256];
    return guarded(err, sizeof(err), [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        launch_wire_emit(c->stream, c->d_kp.as<sift_hip_keypoint>(), c->d_desc.as<float>(), c->total, c->d_wire_off.as<long long>(),
                         static_cast<uint8_t*>(d_records), static_cast<float*>(d_values));
        SIFT_HIP_CHECK(hipGetLastError());
        wait_stream(c, c->stream);
        return SIFT_HIP_OK;
    });
}

// The pack without the wait: the kernel goes to the context's SIDE stream and an event behind it, and the call returns.  The
// caller may start the context's next batch at once - its descriptor stage, the first thing that rewrites the arrays the pack
// reads, waits for that event on the device - and calls sift_hip_result_pack_wait (any thread) before it reads the packed lists.
int sift_hip_result_sparse_pack_async(sift_hip_ctx* c, void* d_records, void* d_values) {
    if (!c || !c->have_result || c->wire_for_total != c->total || c->wire_values < 0) return SIFT_HIP_EINVAL;
    if (c->total > 0 && (!d_records || (c->wire_values > 0 && !d_values))) return SIFT_HIP_EINVAL;
    char err[256];
    return guarded(err, sizeof(err), [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        if (!c->ev_pack) { ApiGuard api; SIFT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming)); }
        launch_wire_emit(c->stream2, c->d_kp.as<sift_hip_keypoint>(), c->d_desc.as<float>(), c->total, c->d_wire_off.as<long long>(),
                         static_cast<uint8_t*>(d_records), static_cast<float*>(d_values));
        SIFT_HIP_CHECK(hipGetLastError());
        SIFT_HIP_CHECK(hipEventRecord(c->ev_pack, c->stream2));
        c->pack_pending = true;
        return SIFT_HIP_OK;
    });
}

int sift_hip_result_pack_wait(sift_hip_ctx* c) {
    if (!c) return SIFT_HIP_EINVAL;
    hipEvent_t ev = c->ev_pack;     // (only the event is touched: the context's own thread may be inside its next batch)
    if (!ev) return SIFT_HIP_OK;
    // Polled with a pause of ~30 us between queries, not in a tight loop: the caller is a gather thread whose lists are two
    // steps behind anyway, and every query takes locks inside the runtime that the launching threads of the same process want
    // (round 5: a gather thread that only waited for packs made a step 0.3 - 0.4 ms LONGER than one that also sent them).
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return SIFT_HIP_OK;
        if (e != hipErrorNotReady) return SIFT_HIP_EHIP;
        std::this_thread::sleep_for(std::chrono::microseconds(30));
    }
}

// The same lists to HOST memory: packed on the GPU, then only the 34-byte records and the floats that are set cross the link
// (~200 instead of 532 bytes per keypoint).  After sift_hip_result_sparse_size; records: total * 34 bytes, values: n_values floats.
int sift_hip_result_copy_sparse(sift_hip_ctx* c, void* records, float* values) {
    if (!c || !c->have_result || c->wire_for_total != c->total || c->wire_values < 0) return SIFT_HIP_EINVAL;
    FirstBatch first_copy(&c->warm_copies, c->device);
    if (c->total > 0 && (!records || (c->wire_values > 0 && !values))) return SIFT_HIP_EINVAL;
    if (c->total == 0) return SIFT_HIP_OK;
    char err[256];
    return guarded(err, sizeof(err), [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        c->d_sparse_rec.ensure((size_t)c->total * 34);
        c->d_sparse_val.ensure((size_t)std::max<long long>(c->wire_values, 1) * sizeof(float));
        launch_wire_emit(c->stream, c->d_kp.as<sift_hip_keypoint>(), c->d_desc.as<float>(), c->total, c->d_wire_off.as<long long>(),
                         c->d_sparse_rec.as<uint8_t>(), c->d_sparse_val.as<float>());
        SIFT_HIP_CHECK(hipGetLastError());
        if (is_pinned(records) && is_pinned(values)) {   // both copies queued behind the kernel, one wait
            SIFT_HIP_CHECK(hipMemcpyAsync(records, c->d_sparse_rec.p, (size_t)c->total * 34, hipMemcpyDefault, c->stream));
            if (c->wire_values > 0) SIFT_HIP_CHECK(hipMemcpyAsync(values, c->d_sparse_val.p, (size_t)c->wire_values * sizeof(float), hipMemcpyDefault, c->stream));
            wait_stream(c, c->stream);
        } else {
            download(c, records, c->d_sparse_rec.p, (size_t)c->total * 34, c->stream);
            if (c->wire_values > 0) download(c, values, c->d_sparse_val.p, (size_t)c->wire_values * sizeof(float), c->stream);
        }
        return SIFT_HIP_OK;
    });
}

// Host side of the format: n records + their floats -> n sift_hip_keypoint and n x 128 descriptor floats, bit for bit what
// sift_hip_result_copy would have delivered.  Plain CPU code; `threads` <= 1 runs on the calling thread.
namespace {
struct UnpackJob {
    const uint8_t* rec; const float* val; int64_t n; sift_hip_keypoint* kp; float* desc;
    std::vector<int64_t> start;   // first value of every chunk
    int64_t chunk;
    const float* vend;            // one past the last value (the vector form reads 8 floats at a time)
};
inline int popcount_mask(const uint8_t* m) {
    uint64_t a, b;
    std::memcpy(&a, m, 8);
    std::memcpy(&b, m + 6, 8);   // bytes 6..13: keep the top 6 bytes (8..13)
    return __builtin_popcountll(a) + __builtin_popcountll(b >> 16);
}
// AVX2 form of one descriptor: a cell's (at most 7) set floats are the next popcount(bits) values; one unaligned 8-float load,
// one lane permutation by a table entry (lane b <- the rank of bit b among the set bits) and one AND with the entry's lane mask
// put them in place - 3 vector instructions per cell instead of a 7-trip loop of tests.  The load may look up to 7 floats past
// the cell's own values, so it is only used while 8 floats remain before `vend`; the tail takes the scalar loop.
struct CellLut {
    alignas(32) int idx[128][8];
    alignas(32) int msk[128][8];
    CellLut() {
        for (int m = 0; m < 128; ++m) {
            int rank = 0;
            for (int b = 0; b < 8; ++b) {
                const bool set = b < 7 && ((m >> b) & 1);
                idx[m][b] = set ? rank : 0;
                msk[m][b] = set ? -1 : 0;
                rank += set ? 1 : 0;
            }
        }
    }
};
inline unsigned cell_bits(const uint8_t* m, int cell) {
    // presence bit cell*7+bin <-> descriptor float cell*8+bin; bin 7 is never on the wire (+0.0f)
    const int bit0 = cell * 7;
    return (unsigned)((m[bit0 >> 3] | (m[(bit0 >> 3) + 1 < 14 ? (bit0 >> 3) + 1 : 13] << 8)) >> (bit0 & 7)) & 0x7fu;
}
__attribute__((target("avx2"))) const float* unpack_desc_avx2(const uint8_t* m, const float* v, const float* vend, float* d, const CellLut& lut) {
    for (int cell = 0; cell < 16; ++cell) {
        const unsigned bits = cell_bits(m, cell);
        float* dc = d + cell * 8;
        if (v + 8 <= vend) {
            const __m256 x = _mm256_loadu_ps(v);
            const __m256 y = _mm256_permutevar8x32_ps(x, _mm256_load_si256(reinterpret_cast<const __m256i*>(lut.idx[bits])));
            _mm256_storeu_ps(dc, _mm256_and_ps(y, _mm256_castsi256_ps(_mm256_load_si256(reinterpret_cast<const __m256i*>(lut.msk[bits])))));
            v += __builtin_popcount(bits);
        } else {
            for (int b = 0; b < 7; ++b) { dc[b] = (bits >> b) & 1u ? *v++ : 0.0f; }
            dc[7] = 0.0f;
        }
    }
    return v;
}
void unpack_chunk(int part, void* arg) {
    const UnpackJob* j = static_cast<const UnpackJob*>(arg);
    const int64_t i0 = (int64_t)part * j->chunk, i1 = std::min(j->n, i0 + j->chunk);
    const float* v = j->val + j->start[(size_t)part];
    static const CellLut lut;
    static const bool avx2 = __builtin_cpu_supports("avx2");
    for (int64_t i = i0; i < i1; ++i) {
        const uint8_t* r = j->rec + i * 34;
        if (j->kp) std::memcpy(&j->kp[i], r, sizeof(sift_hip_keypoint));
        const uint8_t* m = r + 20;
        float* d = j->desc ? j->desc + i * 128 : nullptr;
        if (d && avx2) {
            v = unpack_desc_avx2(m, v, j->vend, d, lut);
            continue;
        }
        for (int cell = 0; cell < 16; ++cell) {
            const unsigned bits = cell_bits(m, cell);
            if (d) {
                float* dc = d + cell * 8;
                for (int b = 0; b < 7; ++b) { dc[b] = (bits >> b) & 1u ? *v++ : 0.0f; }
                dc[7] = 0.0f;
            } else {
                v += __builtin_popcount(bits);
            }
        }
    }
}
}  // namespace

int sift_hip_sparse_unpack_host(const void* records, const float* values, int64_t n_keypoints, sift_hip_keypoint* keypoints,
                                float* descriptors, int threads) {
    if (n_keypoints < 0 || (n_keypoints > 0 && !records)) return SIFT_HIP_EINVAL;
    if (n_keypoints == 0) return SIFT_HIP_OK;
    try {
        UnpackJob j{static_cast<const uint8_t*>(records), values, n_keypoints, keypoints, descriptors, {}, 0, nullptr};
        const int parts = (int)std::min<int64_t>(std::max(threads, 1), (n_keypoints + 1023) / 1024);
        j.chunk = (n_keypoints + parts - 1) / parts;
        j.start.assign((size_t)parts, 0);
        int64_t at = 0;
        for (int p = 0; p < parts; ++p) {   // where every chunk's floats begin: one pass over the presence bits
            j.start[(size_t)p] = at;
            const int64_t i1 = std::min(n_keypoints, (int64_t)(p + 1) * j.chunk);
            for (int64_t i = (int64_t)p * j.chunk; i < i1; ++i) at += popcount_mask(j.rec + i * 34 + 20);
        }
        j.vend = values ? values + at : nullptr;
        parallel_for(parts, parts, unpack_chunk, &j);
        return SIFT_HIP_OK;
    } catch (const std::exception&) {
        return SIFT_HIP_EHIP;
    }
}

int sift_hip_sparse_unpack(sift_hip_ctx* c, const void* d_records, const void* d_values, int64_t n_keypoints, void* d_keypoints,
                           void* d_descriptors) {
    if (!c || n_keypoints < 0 || (n_keypoints > 0 && (!d_records || !d_keypoints || !d_descriptors))) return SIFT_HIP_EINVAL;
    if (n_keypoints == 0) return SIFT_HIP_OK;
    char err[256];
    return guarded(err, sizeof(err), [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        const size_t nb = wire_blocks(n_keypoints);
        c->d_unpack_sums.ensure((nb + 1) * sizeof(int));
        c->d_unpack_off.ensure((nb + 1) * sizeof(long long));
        launch_wire_unpack(c->stream, static_cast<const uint8_t*>(d_records), static_cast<const float*>(d_values), n_keypoints,
                           c->d_unpack_sums.as<int>(), c->d_unpack_off.as<long long>(), static_cast<sift_hip_keypoint*>(d_keypoints),
                           static_cast<float*>(d_descriptors));
        SIFT_HIP_CHECK(hipGetLastError());
        wait_stream(c, c->stream);
        return SIFT_HIP_OK;
    });
}

int sift_hip_image_dims(sift_hip_ctx* c, int* w, int* h) {
    if (!c || !c->plan.valid) return SIFT_HIP_EINVAL;
    const bool up = c->plan.params.subpixel && c->plan.fail_op > 0;
    *w = up ? c->plan.bw : c->plan.in_w;
    *h = up ? c->plan.bh : c->plan.in_h;
    return SIFT_HIP_OK;
}
int sift_hip_image_copy(sift_hip_ctx* c, int image, float* out) {
    if (!c || !c->plan.valid || !c->have_pyramid || image < 0 || image >= c->plan.n) return SIFT_HIP_EINVAL;
    const bool up = c->plan.params.subpixel && c->plan.fail_op > 0;
    if (!up) return SIFT_HIP_EINVAL;  // image unchanged: the caller still has it
    return guarded(nullptr, 0, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        const size_t px = (size_t)c->plan.bw * (size_t)c->plan.bh;
        SIFT_HIP_CHECK(hipMemcpy(out, c->d_base.as<float>() + (size_t)image * px, px * sizeof(float), hipMemcpyDeviceToHost));
        return SIFT_HIP_OK;
    });
}

static float* level_ptr(sift_hip_ctx* c, int kind, int o, int i, int* w, int* h) {
    const Plan& P = c->plan;
    if (!P.valid || o < 0 || o >= P.O) return nullptr;
    const int nl = kind == 1 ? P.D : P.D + 1;
    if (i < 0 || i >= nl) return nullptr;
    *w = P.dev.w[o];
    *h = P.dev.h[o];
    switch (kind) {
        case 0: return P.dev.gauss[o * (P.D + 1) + i];
        case 1: return P.dogs_carved ? P.dev.dog[o * P.D + i] : c->d_tmp2.as<float>();   // not carved: formed on demand (sift_hip_level_copy)
        case 2: return P.dev.mag[o * (P.D + 1) + i];
        case 3: return P.dev.ori[o * (P.D + 1) + i];
    }
    return nullptr;
}
int sift_hip_level_dims(sift_hip_ctx* c, int kind, int octave, int level, int* w, int* h) {
    if (!c) return SIFT_HIP_EINVAL;
    int ww = 0, hh = 0;
    float* p = level_ptr(c, kind, octave, level, &ww, &hh);
    *w = p ? ww : 0;
    *h = p ? hh : 0;
    return p ? SIFT_HIP_OK : SIFT_HIP_EINVAL;
}
int sift_hip_level_copy(sift_hip_ctx* c, int image, int kind, int octave, int level, float* out) {
    if (!c || !c->have_pyramid || image < 0 || image >= c->plan.n) return SIFT_HIP_EINVAL;
    int w = 0, h = 0;
    float* p = level_ptr(c, kind, octave, level, &w, &h);
    if (!p) return SIFT_HIP_EINVAL;
    return guarded(nullptr, 0, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        if (kind == 1 && c->dogs_missing) {
            // the batch wrote no DoG level (the extremum scan forms its tiles from the Gaussian levels) and the plan holds none:
            // this one is formed now, by alg::dog's own kernel, from the two Gaussian levels it lies between, in the scratch image
            const Plan& P = c->plan;
            const int di = octave * P.D + level;
            if (c->dog_in_scratch != di) {
                auto formed = [&](int j) {   // did the batch's pyramid reach g(octave, j)?
                    for (size_t k = 0; k < P.ops.size() && k < P.fail_op; ++k) {
                        const BlurOp& op = P.ops[k];
                        if (j == 0 ? ((octave == 0 && op.kind == 1) || (octave > 0 && op.kind == 3 && op.octave == octave - 1))
                                   : (op.kind == 2 && op.octave == octave && op.j == j))
                            return true;
                    }
                    return false;
                };
                if (!formed(level) || !formed(level + 1)) return (int)SIFT_HIP_EINVAL;   // the pyramid stopped before this level (precondition)
                const int gl = octave * (P.D + 1) + level;
                launch_dog(c->stream, P.dev.gauss[gl], P.dev.gauss[gl + 1], p, (size_t)w * (size_t)h * (size_t)P.n);
                SIFT_HIP_CHECK(hipGetLastError());
                SIFT_HIP_CHECK(hipStreamSynchronize(c->stream));
                c->dog_in_scratch = di;
            }
        }
        const size_t px = (size_t)w * (size_t)h;
        SIFT_HIP_CHECK(hipMemcpy(out, p + (size_t)image * px, px * sizeof(float), hipMemcpyDeviceToHost));
        return SIFT_HIP_OK;
    });
}
float sift_hip_level_scale(sift_hip_ctx* c, int kind, int octave, int level) {
    if (!c || !c->plan.valid || octave < 0 || octave >= c->plan.O || level < 0 || level >= (kind == 1 ? c->plan.D : c->plan.D + 1)) return NAN;
    return kind == 1 ? c->plan.dev.dog_scale[octave * c->plan.D + level] : c->plan.dev.gauss_scale[octave * (c->plan.D + 1) + level];
}

int sift_hip_stage_count(sift_hip_ctx* c, int image, int stage) {
    if (!c || !c->have_result || image < 0 || image >= c->plan.n) return -1;
    if (stage != 4) {
        try {
            SIFT_HIP_CHECK(hipSetDevice(c->device));
            ensure_host_stages(c);
        } catch (...) {
            return -1;
        }
    }
    switch (stage) {
        case 0: return c->totals[(size_t)image];
        case 1: return (int)c->list1[(size_t)image].size();
        case 2: return (int)c->after_orient[(size_t)image].size();
        case 3: return (int)c->final_list[(size_t)image].size();
        case 4: return c->counts[(size_t)image];
    }
    return -1;
}
int sift_hip_stage_copy(sift_hip_ctx* c, int image, int stage, sift_hip_keypoint* out) {
    const int cnt = sift_hip_stage_count(c, image, stage);
    if (cnt < 0) return SIFT_HIP_EINVAL;
    if (cnt == 0) return SIFT_HIP_OK;
    return guarded(nullptr, 0, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        const Plan& P = c->plan;
        if (stage == 4) {
            SIFT_HIP_CHECK(hipMemcpy(out, c->d_kp.as<sift_hip_keypoint>() + c->out_base[(size_t)image],
                                     (size_t)cnt * sizeof(sift_hip_keypoint), hipMemcpyDeviceToHost));
            return SIFT_HIP_OK;
        }
        std::vector<Candidate> hc((size_t)c->totals[(size_t)image]);
        SIFT_HIP_CHECK(hipMemcpy(hc.data(), c->d_cands.as<Candidate>() + (size_t)image * (size_t)P.dev.cand_capacity,
                                 hc.size() * sizeof(Candidate), hipMemcpyDeviceToHost));
        auto fill = [&](sift_hip_keypoint& k, uint32_t cand, float orientation, uint8_t filtered) {
            const Candidate& cd = hc[cand];
            k.scale = P.dev.dog_scale[cd.octave * P.D + cd.index];
            k.orientation = orientation;
            k.x = cd.x; k.y = cd.y; k.octave = cd.octave; k.index = cd.index;
            k.filtered = filtered; k.has_descriptor = 0; k.reserved = 0;
        };
        if (stage == 0) {
            const uint8_t* fl = c->h_flags.as<uint8_t>() + c->flag_off[(size_t)image];
            for (int i = 0; i < cnt; ++i) fill(out[i], (uint32_t)i, 0.0f, fl[i]);
        } else if (stage == 1) {
            const auto& l = c->list1[(size_t)image];
            for (int i = 0; i < cnt; ++i) fill(out[i], l[(size_t)i], 0.0f, 0);
        } else {
            const auto& l = stage == 2 ? c->after_orient[(size_t)image] : c->final_list[(size_t)image];
            for (int i = 0; i < cnt; ++i) fill(out[i], l[(size_t)i].cand, l[(size_t)i].orientation, l[(size_t)i].filtered);
        }
        return SIFT_HIP_OK;
    });
}

// ---- single-image operators ---------------------------------------------------------------------------
int sift_hip_gauss_taps(float sigma, float* taps, int cap) {
    std::vector<float> t;
    int r = 0;
    if (!gauss_taps(sigma, t, r)) return -1;
    for (int i = 0; i < (int)t.size() && i < cap; ++i) taps[i] = t[(size_t)i];
    return r;
}

}  // extern "C"

namespace {
struct Scratch {
    std::vector<void*> ptrs;
    template <class T>
    T* dev(size_t count) {
        void* p = nullptr;
        ApiGuard api;
        SIFT_HIP_CHECK(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
        ptrs.push_back(p);
        return static_cast<T*>(p);
    }
    template <class T>
    T* upload(const T* h, size_t count) {
        T* d = dev<T>(count);
        SIFT_HIP_CHECK(hipMemcpy(d, h, count * sizeof(T), hipMemcpyHostToDevice));
        return d;
    }
    ~Scratch() {
        ApiGuard api;
        for (void* p : ptrs) (void)hipFree(p);
    }
};

int blur_checks(float sigma, int w, int h, std::vector<float>& taps, int& r, std::string& msg) {
    if (!gauss_taps(sigma, taps, r)) { msg = precondition("Kernel1D::initGaussian(): Standard deviation must be >= 0."); return SIFT_HIP_EPRECONDITION; }
    if (!(w >= r + 1)) { msg = precondition("separableConvolveX(): kernel longer than line\n"); return SIFT_HIP_EPRECONDITION; }
    if (!(h >= r + 1)) { msg = precondition("separableConvolveY(): kernel longer than line\n"); return SIFT_HIP_EPRECONDITION; }
    return 0;
}

// blur then optional resample to (wd, hd)
int op_blur_resample(sift_hip_ctx* c, const float* in, int w, int h, float sigma, int wd, int hd, float* out, char* err, int errlen) {
    if (!c || !in || !out || w <= 0 || h <= 0) return SIFT_HIP_EINVAL;
    return guarded(err, errlen, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        std::vector<float> taps;
        int r = 0;
        std::string msg;
        int rc = blur_checks(sigma, w, h, taps, r, msg);
        if (rc) { set_err(err, errlen, msg); return rc; }
        const bool resample = wd > 0;
        if (resample) {
            if (!(w > 1 && h > 1)) { set_err(err, errlen, precondition("resizeImageNoInterpolation(): Source image too small.\n")); return SIFT_HIP_EPRECONDITION; }
            if (!(wd > 1 && hd > 1)) { set_err(err, errlen, precondition("resizeImageNoInterpolation(): Destination image too small.\n")); return SIFT_HIP_EPRECONDITION; }
        }
        Scratch s;
        const size_t px = (size_t)w * (size_t)h;
        float* d_in = s.upload(in, px);
        float* d_tmp = s.dev<float>(px);
        float* d_out = s.dev<float>(px);
        float* d_taps = s.upload(taps.data(), taps.size());
        launch_blur(c->stream, c->fused, d_in, d_tmp, d_out, nullptr, w, h, 1, d_taps, r, min_waves(c));
        if (resample) {
            const std::vector<int> lx = resize_index_map(w, wd), ly = resize_index_map(h, hd);
            int* d_lx = s.upload(lx.data(), lx.size());
            int* d_ly = s.upload(ly.data(), ly.size());
            float* d_rs = s.dev<float>((size_t)wd * (size_t)hd);
            launch_resample(c->stream, d_out, d_rs, w, h, wd, hd, 1, d_lx, d_ly);
            SIFT_HIP_CHECK(hipStreamSynchronize(c->stream));
            SIFT_HIP_CHECK(hipMemcpy(out, d_rs, (size_t)wd * (size_t)hd * sizeof(float), hipMemcpyDeviceToHost));
        } else {
            SIFT_HIP_CHECK(hipStreamSynchronize(c->stream));
            SIFT_HIP_CHECK(hipMemcpy(out, d_out, px * sizeof(float), hipMemcpyDeviceToHost));
        }
        return SIFT_HIP_OK;
    });
}
}  // namespace

extern "C" {

int sift_hip_convolve_with_gauss(sift_hip_ctx* c, const float* in, int w, int h, float sigma, float* out, char* err, int errlen) {
    return op_blur_resample(c, in, w, h, sigma, 0, 0, out, err, errlen);
}
int sift_hip_reduce_to_next_level(sift_hip_ctx* c, const float* in, int w, int h, float sigma, float* out, char* err, int errlen) {
    return op_blur_resample(c, in, w, h, sigma, (w + 1) / 2, (h + 1) / 2, out, err, errlen);
}
int sift_hip_increase_to_next_level(sift_hip_ctx* c, const float* in, int w, int h, float sigma, float* out, char* err, int errlen) {
    return op_blur_resample(c, in, w, h, sigma, 2 * w, 2 * h, out, err, errlen);
}

int sift_hip_dog(sift_hip_ctx* c, const float* lower, const float* higher, int w, int h, float* out) {
    if (!c || !lower || !higher || !out || w <= 0 || h <= 0) return SIFT_HIP_EINVAL;
    return guarded(nullptr, 0, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        Scratch s;
        const size_t px = (size_t)w * (size_t)h;
        float* a = s.upload(lower, px);
        float* b = s.upload(higher, px);
        float* o = s.dev<float>(px);
        launch_dog(c->stream, a, b, o, px);
        SIFT_HIP_CHECK(hipStreamSynchronize(c->stream));
        SIFT_HIP_CHECK(hipMemcpy(out, o, px * sizeof(float), hipMemcpyDeviceToHost));
        return SIFT_HIP_OK;
    });
}

int sift_hip_gradient(sift_hip_ctx* c, const float* in, int w, int h, float* mag, float* ori) {
    if (!c || !in || !mag || !ori || w <= 0 || h <= 0) return SIFT_HIP_EINVAL;
    return guarded(nullptr, 0, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        Scratch s;
        const size_t px = (size_t)w * (size_t)h;
        float* a = s.upload(in, px);
        float* m = s.dev<float>(px);
        float* o = s.dev<float>(px);
        float* pr = s.dev<float>(px);
        launch_gradient(c->stream, a, m, o, pr, w, h, 1);
        SIFT_HIP_CHECK(hipStreamSynchronize(c->stream));
        SIFT_HIP_CHECK(hipMemcpy(mag, m, px * sizeof(float), hipMemcpyDeviceToHost));
        SIFT_HIP_CHECK(hipMemcpy(ori, o, px * sizeof(float), hipMemcpyDeviceToHost));
        return SIFT_HIP_OK;
    });
}

int sift_hip_edge_responses(sift_hip_ctx* c, const float* dog0, const float* dog1, const float* dog2, int w, int h,
                            const uint16_t* xs, const uint16_t* ys, int m, uint8_t* flags) {
    if (!c || !dog0 || !dog1 || !dog2 || !xs || !ys || !flags || w < 3 || h < 3 || m < 0) return SIFT_HIP_EINVAL;
    for (int i = 0; i < m; ++i)
        if (xs[i] < 1 || xs[i] > w - 2 || ys[i] < 1 || ys[i] > h - 2) return SIFT_HIP_EINVAL;
    if (m == 0) return SIFT_HIP_OK;
    return guarded(nullptr, 0, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        Scratch s;
        const size_t px = (size_t)w * (size_t)h;
        float* a = s.upload(dog0, px);
        float* b = s.upload(dog1, px);
        float* d = s.upload(dog2, px);
        uint16_t* dx = s.upload(xs, (size_t)m);
        uint16_t* dy = s.upload(ys, (size_t)m);
        uint8_t* df = s.dev<uint8_t>((size_t)m);
        launch_edge_filter_points(c->stream, a, b, d, w, h, dx, dy, m, df);
        SIFT_HIP_CHECK(hipStreamSynchronize(c->stream));
        SIFT_HIP_CHECK(hipMemcpy(flags, df, (size_t)m, hipMemcpyDeviceToHost));
        return SIFT_HIP_OK;
    });
}

int sift_hip_vertex_parabola(sift_hip_ctx* c, const uint16_t* lnx, const float* lny, const uint16_t* px, const float* py,
                             const uint16_t* rnx, const float* rny, int m, float* out) {
    if (!c || m < 0) return SIFT_HIP_EINVAL;
    if (m == 0) return SIFT_HIP_OK;
    return guarded(nullptr, 0, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        Scratch s;
        uint16_t* a = s.upload(lnx, (size_t)m);
        float* b = s.upload(lny, (size_t)m);
        uint16_t* d = s.upload(px, (size_t)m);
        float* e = s.upload(py, (size_t)m);
        uint16_t* f = s.upload(rnx, (size_t)m);
        float* g = s.upload(rny, (size_t)m);
        float* o = s.dev<float>((size_t)m);
        launch_vertex_parabola(c->stream, a, b, d, e, f, g, m, o);
        SIFT_HIP_CHECK(hipStreamSynchronize(c->stream));
        SIFT_HIP_CHECK(hipMemcpy(out, o, (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
        return SIFT_HIP_OK;
    });
}

int sift_hip_sort_by_filter(sift_hip_ctx* c, const uint8_t* flags, int n, int32_t* perm) {
    (void)c;
    if (!flags || !perm || n < 0) return SIFT_HIP_EINVAL;
    std::vector<uint32_t> p;
    sort_by_filter(flags, n, p);
    for (int i = 0; i < n; ++i) perm[i] = (int32_t)p[(size_t)i];
    return SIFT_HIP_OK;
}

int sift_hip_cleanup_survivors(sift_hip_ctx* c, const uint8_t* flags, int n, int32_t* survivors, int32_t* count, int on_gpu) {
    if (!flags || !survivors || !count || n < 0) return SIFT_HIP_EINVAL;
    if (!on_gpu) {
        std::vector<uint32_t> sv;
        cleanup_survivors(flags, n, sv);
        *count = (int32_t)sv.size();
        for (size_t i = 0; i < sv.size(); ++i) survivors[i] = (int32_t)sv[i];
        return SIFT_HIP_OK;
    }
    if (!c) return SIFT_HIP_EINVAL;
    return guarded(nullptr, 0, [&]() {
        SIFT_HIP_CHECK(hipSetDevice(c->device));
        Scratch s;
        const size_t m = (size_t)std::max(n, 1) + 64;
        uint8_t* d_fl = s.dev<uint8_t>(m);
        if (n) SIFT_HIP_CHECK(hipMemcpy(d_fl, flags, (size_t)n, hipMemcpyHostToDevice));
        uint8_t* wk = s.dev<uint8_t>(m);
        uint32_t* wi = s.dev<uint32_t>(m);
        uint32_t* wi2 = s.dev<uint32_t>(m);
        uint32_t* wp = s.dev<uint32_t>(m);
        uint32_t* out = s.dev<uint32_t>(m);
        int* info = s.dev<int>(2);
        unsigned long long* st = nullptr;
        if (c->diag_cleanup_stamps) {
            st = s.dev<unsigned long long>(512);
            SIFT_HIP_CHECK(hipMemset(st, 0, 512 * sizeof(unsigned long long)));
            cleanup_set_stamp_buffer(st);
        }
        OrientIn* kord = s.dev<OrientIn>(65536);
        uint32_t* klr = s.dev<uint32_t>(65536);
        Candidate* kcd = s.dev<Candidate>(m);
        SIFT_HIP_CHECK(hipMemset(kcd, 0, m * sizeof(Candidate)));
        launch_cleanup_kat(c->stream, d_fl, n, wk, wi, wi2, wp, out, info, on_gpu == 2 ? 1 : 0, kord, klr, kcd);
        SIFT_HIP_CHECK(hipStreamSynchronize(c->stream));
        int h_info[2];
        SIFT_HIP_CHECK(hipMemcpy(h_info, info, sizeof(h_info), hipMemcpyDeviceToHost));
        if (st) {
            unsigned long long hs[512];
            SIFT_HIP_CHECK(hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost));
            if (hs[7]) {   // per-round stamps of the word-parallel rounds: start, after pivot step, after scan, after swaps
                std::fprintf(stderr, "rounds %llu:", hs[7]);
                for (unsigned long long r = 0; r < hs[7] && r < 120; ++r)
                    std::fprintf(stderr, " [p%llu n%llu %.1f %.1f %.1f %.1f]", hs[16 + 4 * r + 3] >> 60, (hs[16 + 4 * r + 3] >> 32) & 0xfffffff,
                                 (hs[16 + 4 * r + 1] - hs[16 + 4 * r]) / 100.0, (hs[16 + 4 * r + 2] - hs[16 + 4 * r + 1]) / 100.0,
                                 ((hs[16 + 4 * r + 3] & 0xffffffffull) - (hs[16 + 4 * r + 2] & 0xffffffffull)) / 100.0,
                                 r + 1 < hs[7] ? (hs[16 + 4 * (r + 1)] - hs[16 + 4 * r]) / 100.0 : 0.0);
                std::fprintf(stderr, "\n");
            }
            cleanup_set_stamp_buffer(nullptr);
            if (hs[8])
                std::fprintf(stderr, "cleanup1 stamps (us): bits %.1f ranks %.1f loop %.1f copy %.1f pure %.1f compact %.1f  npure %llu\n",
                             (hs[8] - hs[0]) / 100.0, (hs[9] - hs[8]) / 100.0, (hs[2] - hs[1]) / 100.0, (hs[3] - hs[2]) / 100.0,
                             (hs[4] - hs[3]) / 100.0, (hs[5] - hs[4]) / 100.0, hs[6]);
            else
                std::fprintf(stderr, "cleanup stamps (us): init %.1f loop %.1f copy %.1f pure %.1f compact %.1f  npure %llu\n",
                             (hs[1] - hs[0]) / 100.0, (hs[2] - hs[1]) / 100.0, (hs[3] - hs[2]) / 100.0, (hs[4] - hs[3]) / 100.0,
                             (hs[5] - hs[4]) / 100.0, hs[6]);
        }
        if (h_info[1]) {  // introsort depth limit: the host's std::sort decides
            std::vector<uint32_t> sv;
            cleanup_survivors(flags, n, sv);
            *count = (int32_t)sv.size();
            for (size_t i = 0; i < sv.size(); ++i) survivors[i] = (int32_t)sv[i];
            return SIFT_HIP_OK;
        }
        *count = h_info[0];
        if (h_info[0]) SIFT_HIP_CHECK(hipMemcpy(survivors, out, (size_t)h_info[0] * sizeof(uint32_t), hipMemcpyDeviceToHost));
        return SIFT_HIP_OK;
    });
}

int sift_hip_profile_get(sift_hip_ctx* c, int which, double* ms, int64_t* launches, double* bytes) {
    if (!c || which < 0 || which >= sift_hip_ctx::kProfClasses) return SIFT_HIP_EINVAL;
    if (ms) *ms = c->prof_ms[which];
    if (launches) *launches = c->prof_launches[which];
    if (bytes) *bytes = c->prof_bytes[which];
    return SIFT_HIP_OK;
}

int sift_hip_profile_get_busy(sift_hip_ctx* c, int which, double* busy_ms) {
    if (!c || which < 0 || which >= sift_hip_ctx::kProfClasses || !busy_ms) return SIFT_HIP_EINVAL;
    *busy_ms = c->prof_busy_ms[which];
    return SIFT_HIP_OK;
}
int sift_hip_profile_reset(sift_hip_ctx* c) {
    if (!c) return SIFT_HIP_EINVAL;
    for (int i = 0; i < sift_hip_ctx::kProfClasses; ++i) { c->prof_ms[i] = 0; c->prof_launches[i] = 0; c->prof_bytes[i] = 0; c->prof_busy_ms[i] = 0; }
    c->prof_batches = 0;
    return SIFT_HIP_OK;
}
int sift_hip_profile_batches(sift_hip_ctx* c, int64_t* batches) {
    if (!c || !batches) return SIFT_HIP_EINVAL;
    *batches = c->prof_batches;
    return SIFT_HIP_OK;
}

}  // extern "C"
