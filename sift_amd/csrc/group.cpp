// Several GPUs of one node from ONE process, no torch: SURVEY.md 8(e)'s layout as native host code.  A batch of independent
// frames (sift.cpp keeps no state between images) is block-sharded over the shards of a group; every shard is one
// sift_hip_ctx driven by its own host thread; nothing but keypoint lists crosses devices: the per-shard result arrays
// (20-byte records, 128-float descriptors) are copied device-to-device (hipMemcpyPeerAsync: xGMI between GPUs, each shard
// over its own link to shard 0's GPU) into one array in global image order on shard 0's device.  No collective: shards never
// wait for each other except at this gather.  Written against the public C ABI only (include/sift_hip.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sift_hip.h"

struct sift_hip_group {
    std::vector<int> devices;
    std::vector<sift_hip_ctx*> ctx;
    // last batch
    int n = 0;
    std::vector<int> first, count;           // frames of every shard: [first, first + count)
    std::vector<int> rc;
    std::vector<std::string> msg;
    std::vector<int32_t> status, counts;     // per image, global order
    long long total = 0;
    bool have_result = false;
    void* d_kp = nullptr;                    // on devices[0]
    void* d_desc = nullptr;
    long long cap = 0;
    hipStream_t copy_stream = nullptr;       // on devices[0]
    double gather_ms = 0, compute_ms = 0;
    long long gather_bytes = 0;
    // option "gather_wire": 1 (default) lists of other GPUs cross in the sparse wire format (34-byte records + the descriptor
    // floats that are set: ~200 instead of 532 bytes per keypoint over the link) and are unpacked on devices[0]; 0 plain arrays;
    // 2 the sparse format for every shard, also those on devices[0] itself (tests on a one-GPU box)
    int gather_wire = 1;
    std::vector<void*> s_rec, s_val;         // per shard, on the shard's device: packed records / values
    std::vector<long long> s_rec_cap, s_val_cap, s_nnz;
    std::vector<int> s_packed;               // this batch: the shard's lists are packed (lossless and wanted)
    void* d_in_rec = nullptr;                // on devices[0]: where packed lists arrive
    void* d_in_val = nullptr;
    long long in_rec_cap = 0, in_val_cap = 0;
};

namespace {
void set_err(char* err, int errlen, const std::string& m) {
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", m.c_str());
}
double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

extern "C" {

int sift_hip_group_create(const int* devices, int n_devices, sift_hip_group** out, char* err, int errlen) {
    if (!devices || n_devices <= 0 || !out) return SIFT_HIP_EINVAL;
    *out = nullptr;
    auto* g = new sift_hip_group();
    g->devices.assign(devices, devices + n_devices);
    for (int s = 0; s < n_devices; ++s) {
        sift_hip_ctx* c = nullptr;
        const int rc = sift_hip_create(devices[s], &c, err, errlen);
        if (rc != SIFT_HIP_OK) {
            for (auto* p : g->ctx) sift_hip_destroy(p);
            delete g;
            return rc;
        }
        g->ctx.push_back(c);
    }
    // direct peer copies into shard 0's device (already-enabled and same-device answers are fine)
    for (int s = 1; s < n_devices; ++s)
        if (devices[s] != devices[0]) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[s], devices[0]) == hipSuccess && can) {
                (void)hipSetDevice(devices[s]);
                (void)hipDeviceEnablePeerAccess(devices[0], 0);
            }
            (void)hipGetLastError();
        }
    (void)hipSetDevice(devices[0]);
    if (hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        set_err(err, errlen, "sift_hip_group_create: cannot create the gather stream");
        for (auto* p : g->ctx) sift_hip_destroy(p);
        delete g;
        return SIFT_HIP_EHIP;
    }
    *out = g;
    return SIFT_HIP_OK;
}

void sift_hip_group_destroy(sift_hip_group* g) {
    if (!g) return;
    (void)hipSetDevice(g->devices[0]);
    if (g->copy_stream) {
        (void)hipStreamSynchronize(g->copy_stream);
        (void)hipStreamDestroy(g->copy_stream);
    }
    if (g->d_kp) (void)hipFree(g->d_kp);
    if (g->d_desc) (void)hipFree(g->d_desc);
    if (g->d_in_rec) (void)hipFree(g->d_in_rec);
    if (g->d_in_val) (void)hipFree(g->d_in_val);
    for (size_t s = 0; s < g->s_rec.size(); ++s) {
        (void)hipSetDevice(g->devices[s]);
        if (g->s_rec[s]) (void)hipFree(g->s_rec[s]);
        if (g->s_val[s]) (void)hipFree(g->s_val[s]);
    }
    for (auto* c : g->ctx) sift_hip_destroy(c);
    delete g;
}

int sift_hip_group_shards(sift_hip_group* g) { return g ? (int)g->ctx.size() : -1; }

int sift_hip_group_set_option(sift_hip_group* g, const char* name, int value) {
    if (!g || !name) return SIFT_HIP_EINVAL;
    if (!std::strcmp(name, "gather_wire")) {
        if (value < 0 || value > 2) return SIFT_HIP_EINVAL;
        g->gather_wire = value;
        return SIFT_HIP_OK;
    }
    int rc = SIFT_HIP_OK;
    for (auto* c : g->ctx) rc = std::max(rc, sift_hip_set_option(c, name, value));
    return rc;
}

static int group_calculate(sift_hip_group* g, const float* host_imgs, int n, int w, int h, const sift_hip_params* params, char* err, int errlen);

int sift_hip_group_calculate(sift_hip_group* g, const float* host_imgs, int n, int w, int h, const sift_hip_params* params,
                             char* err, int errlen) {
    try {   // no C++ exception (thread creation, allocation) leaves the C ABI
        return group_calculate(g, host_imgs, n, w, h, params, err, errlen);
    } catch (const std::exception& e) {
        set_err(err, errlen, std::string("sift_hip_group_calculate: ") + e.what());
        return SIFT_HIP_EHIP;
    }
}

static int group_calculate(sift_hip_group* g, const float* host_imgs, int n, int w, int h, const sift_hip_params* params, char* err, int errlen) {
    if (!g || !host_imgs || !params || n <= 0 || w <= 0 || h <= 0) return SIFT_HIP_EINVAL;
    const int S = (int)g->ctx.size();
    g->have_result = false;
    g->n = n;
    g->first.assign((size_t)S, 0);
    g->count.assign((size_t)S, 0);
    g->rc.assign((size_t)S, SIFT_HIP_OK);
    g->msg.assign((size_t)S, std::string());
    g->s_rec.resize((size_t)S, nullptr); g->s_val.resize((size_t)S, nullptr);
    g->s_rec_cap.resize((size_t)S, 0); g->s_val_cap.resize((size_t)S, 0);
    g->s_nnz.assign((size_t)S, 0); g->s_packed.assign((size_t)S, 0);
    if (g->gather_wire) sift_hip_group_set_option(g, "wire_count", 1);   // the descriptor kernels count what the wire will carry
    const int per = (n + S - 1) / S;   // contiguous blocks: shard s holds images s*per .. (256 -> 32 each on 8 GPUs)
    for (int s = 0; s < S; ++s) {
        g->first[(size_t)s] = std::min(n, s * per);
        g->count[(size_t)s] = std::min(n, (s + 1) * per) - g->first[(size_t)s];
    }
    const size_t frame = (size_t)w * (size_t)h;
    const double t0 = now_ms();
    struct Joiner {
        std::vector<std::thread> th;
        ~Joiner() { for (auto& t : th) if (t.joinable()) t.join(); }
    } threads;
    std::vector<std::thread>& th = threads.th;
    for (int s = 0; s < S; ++s) {
        if (g->count[(size_t)s] == 0) continue;
        th.emplace_back([g, s, host_imgs, frame, w, h, params]() {
            char e[512] = "";
            g->rc[(size_t)s] = sift_hip_calculate_batch(g->ctx[(size_t)s], host_imgs + (size_t)g->first[(size_t)s] * frame, g->count[(size_t)s], w, h,
                                                       params, e, sizeof(e));
            g->msg[(size_t)s] = e;
            // lists that will cross to another GPU are packed here, on the shard's own GPU and thread
            g->s_packed[(size_t)s] = 0;
            const bool remote = g->devices[(size_t)s] != g->devices[0];
            if ((g->gather_wire == 2 || (g->gather_wire == 1 && remote)) && sift_hip_result_images(g->ctx[(size_t)s]) == g->count[(size_t)s]) {
                const long long t = sift_hip_result_total(g->ctx[(size_t)s]);
                int64_t nnz = 0;
                int lossless = 0;
                if (t > 0 && sift_hip_result_sparse_size(g->ctx[(size_t)s], &nnz, &lossless) == SIFT_HIP_OK && lossless &&
                    hipSetDevice(g->devices[(size_t)s]) == hipSuccess) {
                    auto fit = [](void*& p, long long& cap, long long want) {
                        if (want <= cap) return true;
                        if (p) (void)hipFree(p);
                        p = nullptr; cap = 0;
                        if (hipMalloc(&p, (size_t)(want + want / 4)) != hipSuccess) { p = nullptr; return false; }
                        cap = want + want / 4;
                        return true;
                    };
                    if (fit(g->s_rec[(size_t)s], g->s_rec_cap[(size_t)s], t * 34) &&
                        fit(g->s_val[(size_t)s], g->s_val_cap[(size_t)s], std::max<long long>(nnz, 1) * 4) &&
                        sift_hip_result_sparse_pack(g->ctx[(size_t)s], g->s_rec[(size_t)s], g->s_val[(size_t)s]) == SIFT_HIP_OK) {
                        g->s_nnz[(size_t)s] = nnz;
                        g->s_packed[(size_t)s] = 1;
                    }
                }
            }
        });
    }
    for (auto& t : th) t.join();
    g->compute_ms = now_ms() - t0;

    // per-image status / counts in global order; a shard whose call failed before it ran (bad arguments, HIP error) fails the batch
    g->status.assign((size_t)n, 0);
    g->counts.assign((size_t)n, 0);
    int first_rc = SIFT_HIP_OK;
    std::string first_msg;
    std::vector<long long> shard_total((size_t)S, 0), shard_off((size_t)S, 0);
    long long total = 0;
    for (int s = 0; s < S; ++s) {
        const int cnt = g->count[(size_t)s];
        if (cnt == 0) continue;
        const int rc = g->rc[(size_t)s];
        if (rc != SIFT_HIP_OK && first_rc == SIFT_HIP_OK) { first_rc = rc; first_msg = g->msg[(size_t)s]; }
        if (sift_hip_result_images(g->ctx[(size_t)s]) != cnt) {   // nothing ran on this shard
            if (rc == SIFT_HIP_OK) { first_rc = SIFT_HIP_EHIP; first_msg = "sift_hip_group_calculate: a shard returned no results"; }
            set_err(err, errlen, first_msg);
            return first_rc;
        }
        (void)sift_hip_result_status(g->ctx[(size_t)s], g->status.data() + g->first[(size_t)s], cnt);
        (void)sift_hip_result_counts(g->ctx[(size_t)s], g->counts.data() + g->first[(size_t)s], cnt);
        shard_total[(size_t)s] = sift_hip_result_total(g->ctx[(size_t)s]);
        shard_off[(size_t)s] = total;
        total += shard_total[(size_t)s];
    }
    g->total = total;

    // ---- gather: keypoint lists only, device to device, global image order ------------------------------------
    const double t1 = now_ms();
    if (hipSetDevice(g->devices[0]) != hipSuccess) { set_err(err, errlen, "hipSetDevice failed"); return SIFT_HIP_EHIP; }
    if (total > g->cap) {
        if (g->d_kp) (void)hipFree(g->d_kp);
        if (g->d_desc) (void)hipFree(g->d_desc);
        g->d_kp = g->d_desc = nullptr;
        g->cap = 0;
        const long long want = total + total / 4;
        if (hipMalloc(&g->d_kp, (size_t)want * sizeof(sift_hip_keypoint)) != hipSuccess ||
            hipMalloc(&g->d_desc, (size_t)want * 128 * sizeof(float)) != hipSuccess) {
            set_err(err, errlen, "sift_hip_group_calculate: out of device memory for the gathered lists");
            return SIFT_HIP_EHIP;
        }
        g->cap = want;
    }
    g->gather_bytes = 0;
    // packed lists first: all of them into one staging area on devices[0] (each over its own link), then unpacked there
    {
        long long need_rec = 0, need_val = 0;
        for (int s = 0; s < S; ++s)
            if (g->s_packed[(size_t)s] && shard_total[(size_t)s] > 0) { need_rec += shard_total[(size_t)s] * 34; need_val += g->s_nnz[(size_t)s] * 4; }
        auto fit0 = [](void*& p, long long& cap, long long want) {
            if (want <= cap) return true;
            if (p) (void)hipFree(p);
            p = nullptr; cap = 0;
            if (hipMalloc(&p, (size_t)(want + want / 4)) != hipSuccess) { p = nullptr; return false; }
            cap = want + want / 4;
            return true;
        };
        if (need_rec > 0 && (!fit0(g->d_in_rec, g->in_rec_cap, need_rec) || !fit0(g->d_in_val, g->in_val_cap, std::max<long long>(need_val, 4)))) {
            set_err(err, errlen, "sift_hip_group_calculate: out of device memory for the arriving lists");
            return SIFT_HIP_EHIP;
        }
        long long ro = 0, vo = 0;
        std::vector<long long> rec_at((size_t)S, 0), val_at((size_t)S, 0);
        for (int s = 0; s < S; ++s) {
            if (!g->s_packed[(size_t)s] || shard_total[(size_t)s] <= 0) continue;
            const size_t br = (size_t)shard_total[(size_t)s] * 34, bv = (size_t)g->s_nnz[(size_t)s] * 4;
            char* dr = static_cast<char*>(g->d_in_rec) + ro;
            char* dv = static_cast<char*>(g->d_in_val) + vo;
            rec_at[(size_t)s] = ro; val_at[(size_t)s] = vo;
            ro += (long long)br; vo += (long long)bv;
            hipError_t e1 = hipSuccess, e2 = hipSuccess;
            if (g->devices[(size_t)s] == g->devices[0]) {
                e1 = hipMemcpyAsync(dr, g->s_rec[(size_t)s], br, hipMemcpyDeviceToDevice, g->copy_stream);
                if (bv) e2 = hipMemcpyAsync(dv, g->s_val[(size_t)s], bv, hipMemcpyDeviceToDevice, g->copy_stream);
            } else {
                e1 = hipMemcpyPeerAsync(dr, g->devices[0], g->s_rec[(size_t)s], g->devices[(size_t)s], br, g->copy_stream);
                if (bv) e2 = hipMemcpyPeerAsync(dv, g->devices[0], g->s_val[(size_t)s], g->devices[(size_t)s], bv, g->copy_stream);
                g->gather_bytes += (long long)(br + bv);
            }
            if (e1 != hipSuccess || e2 != hipSuccess) { set_err(err, errlen, "sift_hip_group_calculate: peer copy failed"); return SIFT_HIP_EHIP; }
        }
        if (need_rec > 0) {
            if (hipStreamSynchronize(g->copy_stream) != hipSuccess) { set_err(err, errlen, "sift_hip_group_calculate: gather failed"); return SIFT_HIP_EHIP; }
            for (int s = 0; s < S; ++s) {
                if (!g->s_packed[(size_t)s] || shard_total[(size_t)s] <= 0) continue;
                char* dk = static_cast<char*>(g->d_kp) + (size_t)shard_off[(size_t)s] * sizeof(sift_hip_keypoint);
                char* dd = static_cast<char*>(g->d_desc) + (size_t)shard_off[(size_t)s] * 128 * sizeof(float);
                if (sift_hip_sparse_unpack(g->ctx[0], static_cast<char*>(g->d_in_rec) + rec_at[(size_t)s], static_cast<char*>(g->d_in_val) + val_at[(size_t)s],
                                           shard_total[(size_t)s], dk, dd) != SIFT_HIP_OK) {
                    set_err(err, errlen, "sift_hip_group_calculate: unpacking the arriving lists failed");
                    return SIFT_HIP_EHIP;
                }
            }
        }
    }
    for (int s = 0; s < S; ++s) {
        const long long t = shard_total[(size_t)s];
        if (t <= 0 || g->s_packed[(size_t)s]) continue;
        const void *kp = nullptr, *desc = nullptr;
        if (sift_hip_result_device(g->ctx[(size_t)s], &kp, &desc) != SIFT_HIP_OK) { set_err(err, errlen, "no device results"); return SIFT_HIP_EHIP; }
        char* dk = static_cast<char*>(g->d_kp) + (size_t)shard_off[(size_t)s] * sizeof(sift_hip_keypoint);
        char* dd = static_cast<char*>(g->d_desc) + (size_t)shard_off[(size_t)s] * 128 * sizeof(float);
        const size_t bk = (size_t)t * sizeof(sift_hip_keypoint), bd = (size_t)t * 128 * sizeof(float);
        hipError_t e1, e2;
        if (g->devices[(size_t)s] == g->devices[0]) {
            e1 = hipMemcpyAsync(dk, kp, bk, hipMemcpyDeviceToDevice, g->copy_stream);
            e2 = hipMemcpyAsync(dd, desc, bd, hipMemcpyDeviceToDevice, g->copy_stream);
        } else {
            e1 = hipMemcpyPeerAsync(dk, g->devices[0], kp, g->devices[(size_t)s], bk, g->copy_stream);
            e2 = hipMemcpyPeerAsync(dd, g->devices[0], desc, g->devices[(size_t)s], bd, g->copy_stream);
            g->gather_bytes += (long long)(bk + bd);
        }
        if (e1 != hipSuccess || e2 != hipSuccess) { set_err(err, errlen, "sift_hip_group_calculate: peer copy failed"); return SIFT_HIP_EHIP; }
    }
    if (hipStreamSynchronize(g->copy_stream) != hipSuccess) { set_err(err, errlen, "sift_hip_group_calculate: gather failed"); return SIFT_HIP_EHIP; }
    g->gather_ms = now_ms() - t1;
    g->have_result = true;
    if (first_rc != SIFT_HIP_OK) set_err(err, errlen, first_msg);
    return first_rc;
}

int sift_hip_group_result_images(sift_hip_group* g) { return (g && g->have_result) ? g->n : -1; }
int64_t sift_hip_group_result_total(sift_hip_group* g) { return (g && g->have_result) ? g->total : -1; }
int sift_hip_group_result_status(sift_hip_group* g, int32_t* status, int cap) {
    if (!g || !g->have_result || !status || cap < g->n) return SIFT_HIP_EINVAL;
    std::copy(g->status.begin(), g->status.end(), status);
    return SIFT_HIP_OK;
}
int sift_hip_group_result_counts(sift_hip_group* g, int32_t* counts, int cap) {
    if (!g || !g->have_result || !counts || cap < g->n) return SIFT_HIP_EINVAL;
    std::copy(g->counts.begin(), g->counts.end(), counts);
    return SIFT_HIP_OK;
}
int sift_hip_group_result_device(sift_hip_group* g, const void** dev_keypoints, const void** dev_descriptors) {
    if (!g || !g->have_result) return SIFT_HIP_EINVAL;
    if (dev_keypoints) *dev_keypoints = g->d_kp;
    if (dev_descriptors) *dev_descriptors = g->d_desc;
    return SIFT_HIP_OK;
}
int sift_hip_group_result_copy(sift_hip_group* g, sift_hip_keypoint* kp, float* desc) {
    if (!g || !g->have_result) return SIFT_HIP_EINVAL;
    if (g->total <= 0) return SIFT_HIP_OK;
    if (hipSetDevice(g->devices[0]) != hipSuccess) return SIFT_HIP_EHIP;
    if (kp && hipMemcpyAsync(kp, g->d_kp, (size_t)g->total * sizeof(sift_hip_keypoint), hipMemcpyDefault, g->copy_stream) != hipSuccess) return SIFT_HIP_EHIP;
    if (desc && hipMemcpyAsync(desc, g->d_desc, (size_t)g->total * 128 * sizeof(float), hipMemcpyDefault, g->copy_stream) != hipSuccess) return SIFT_HIP_EHIP;
    return hipStreamSynchronize(g->copy_stream) == hipSuccess ? SIFT_HIP_OK : SIFT_HIP_EHIP;
}
int sift_hip_group_timing(sift_hip_group* g, double* compute_ms, double* gather_ms, int64_t* gather_bytes) {
    if (!g || !g->have_result) return SIFT_HIP_EINVAL;
    if (compute_ms) *compute_ms = g->compute_ms;
    if (gather_ms) *gather_ms = g->gather_ms;
    if (gather_bytes) *gather_bytes = g->gather_bytes;
    return SIFT_HIP_OK;
}

}  // extern "C"
