// Several GPUs of one node from ONE process, no torch: SURVEY.md 8(e)'s layout as native host code.  A batch of independent
// frames (sift.cpp keeps no state between images) is block-sharded over the shards of a group; every shard is one
// sift_hip_ctx driven by its own persistent host thread; nothing but keypoint lists crosses devices.
//
//   shard thread s:   calculate its block  ->  pack its lists into the sparse wire format on its own GPU (34-byte records +
//                     the descriptor floats that are set, ~200 instead of 532 bytes per keypoint)  ->  send them to the first
//                     GPU on its own stream, at once (RCCL ncclSend over its own xGMI link; or a peer copy)  ->  next batch
//   gather thread:    (on devices[0]) once every shard of the batch has reported its sizes: ONE RCCL group with all the
//                     receives (the links run side by side), then unpacks every shard's lists into one array in global image
//                     order (sift_hip_sparse_unpack on a context of its own)
//
// sift_hip_group_submit / _collect keep two batches in flight: the gather of batch k runs under the kernels of batch k+1
// (pack buffers, arrival areas and result arrays are double-buffered).  There is no collective and no step in which shards
// wait for each other except that gather.
//
// Transport.  RCCL (librccl.so.1, opened at run time so that the library also loads where RCCL is absent; one communicator
// per GPU from ncclCommInitAll) when the group's devices are all different, which is the multi-GPU case; RCCL refuses two
// ranks on one GPU, so a group that lists a device twice (tests on a one-GPU box) uses peer / same-device copies, each shard's
// on a stream of its own.  Those copies are kernels of this library (kernels_io.hip: the sending GPU writes the arrival area in
// place, its own memory or the first GPU's through peer access), not hipMemcpyAsync / hipMemcpyPeerAsync: with the runtime's
// device-to-device copies issued by one host thread while another was launching kernels, the LAUNCHES crashed inside the
// runtime (SEGV below hipLaunchKernel in 3 - 8 % of the runs of examples/sift_multi_gpu.cpp; 1 in 200 since, tools/
// example_loop.sh; option "copy_kernels" = 0 brings the runtime's copies back).  For the same reason the caller's ordinary
// memory is reached through page-locked staging buffers, never handed to the runtime as it is.
// Option "gather_transport": 0 copies, 1 RCCL if possible (default), 2 RCCL required.  A group of ONE
// shard with "gather_loopback" = 1 sends its lists through RCCL to itself (ncclSend + ncclRecv to the same rank in one group):
// the whole RCCL path - communicator, grouped point-to-point, arrival area, unpack - on a box with one GPU.
// Written against the public C ABI only (include/sift_hip.h).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sift_hip.h"

extern "C" int sift_hip_internal_copy(void* stream, const void* src, void* dst, size_t bytes);   // kernels_io.hip

#include "launch_guard.h"   // per-device launch locks: allocations and stream / event creation never run beside another thread's launch on that device

namespace {

using ApiGuard = sift_hip::LaunchGuard;
// the runtime's own kernels and markers (copies, event records) go on their streams under the same lock (common.h)
#define hipSetDevice(...) ((hipError_t)sift_hip::set_device_tracked(__VA_ARGS__))
#define hipMemcpyAsync(...) (ApiGuard{}, (hipMemcpyAsync)(__VA_ARGS__))
#define hipMemcpyPeerAsync(...) (ApiGuard{}, (hipMemcpyPeerAsync)(__VA_ARGS__))
#define hipMemcpy(...) (ApiGuard{}, (hipMemcpy)(__VA_ARGS__))
#define hipEventRecord(...) (ApiGuard{}, (hipEventRecord)(__VA_ARGS__))

void set_err(char* err, int errlen, const std::string& m) {
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", m.c_str());
}
double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- RCCL, resolved at run time ------------------------------------------------------------------------------------
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        // a copy the process already holds (PyTorch-ROCm ships one) is found by its SONAME; otherwise the system's
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.lib, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.lib, "ncclRecv"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.GetErrorString;
    });
    return r;
}

struct DevMem {   // grow-only device buffer on a fixed device
    void* p = nullptr;
    long long cap = 0;
    bool fit(int device, long long want) {
        if (want <= cap) return true;
        ApiGuard api(device);
        if (hipSetDevice(device) != hipSuccess) return false;
        // hipFree waits for the whole device - also for an RCCL send of this GPU that is waiting for its receive.  The lock held
        // meanwhile is THIS device's only (launch_guard.h): the other shards and the gather thread get on and post that receive.
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const long long take = want + want / 4 + 256;
        if (hipMalloc(&p, (size_t)take) != hipSuccess) { p = nullptr; (void)hipGetLastError(); return false; }
        cap = take;
        return true;
    }
    void release(int device) {
        if (!p) return;
        ApiGuard api(device);
        (void)hipSetDevice(device);
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

constexpr int kSlots = 2;   // batches in flight

struct ShardBatch {          // what shard s reports for one batch
    int rc = SIFT_HIP_OK;
    std::string msg;
    bool ran = false;        // the context holds results for its whole block
    bool packed = false;     // its lists travel in the sparse wire format
    bool sent = false;       // the shard thread has queued its transfer (copies: the data is under way; RCCL: the sends are posted)
    long long total = 0, nnz = 0;
    double compute_ms = 0;
    std::vector<int32_t> status, counts;
};

struct Batch {
    long long seq = -1;
    int n = 0, w = 0, h = 0;
    const float* imgs = nullptr;
    sift_hip_params params{};
    std::vector<int> first, count;
    std::vector<ShardBatch> shard;
    int reported = 0;        // shards that have finished computing (and queued their transfer)
    bool gathered = false;   // the gather thread is done with it
    int rc = SIFT_HIP_OK;
    std::string msg;
    long long total = 0;
    std::vector<int32_t> status, counts;
    double t_submit = 0, t_computed = 0, t_gathered = 0;
    long long wire_bytes = 0;
};

}  // namespace

struct sift_hip_group {
    std::vector<int> devices;
    std::vector<char> peer_ok;                // per shard: its GPU can write devices[0]'s memory in place (same device, or peer access enabled)
    std::vector<sift_hip_ctx*> ctx;
    sift_hip_ctx* unpack_ctx = nullptr;       // on devices[0]: the gather thread's own (sift_hip_sparse_unpack beside shard 0's kernels)
    int S = 0;
    // options
    int gather_wire = 1;        // 1 (default): lists of other GPUs cross in the sparse wire format; 0 plain arrays; 2 sparse for every shard
    int gather_transport = 1;   // 0 copies, 1 RCCL when the devices allow it, 2 RCCL or fail
    int gather_loopback = 0;    // one shard: its lists go through RCCL to itself
    int copy_kernels = 1;       // device-to-device copies of the gather as kernels of the library (0: hipMemcpyAsync / hipMemcpyPeerAsync)
    // RCCL
    std::vector<ncclComm_t> comm;
    bool comm_tried = false, use_rccl = false;
    std::string rccl_note;
    // threads
    std::mutex m;
    std::condition_variable cv;
    bool stop = false;
    std::vector<std::thread> workers;
    std::thread gatherer;
    long long submitted = 0, collected = 0;     // batches
    std::vector<long long> shard_next;          // next batch each shard thread takes
    long long gather_next = 0;
    Batch batch[kSlots];
    // device memory
    std::vector<hipStream_t> send_stream;       // per shard, on the shard's device
    std::vector<hipEvent_t> sent_ev[kSlots];    // per shard: its transfer of the slot's batch is complete (source side)
    std::vector<DevMem> pack_rec[kSlots], pack_val[kSlots];   // per shard, on the shard's device
    std::vector<DevMem> in_rec[kSlots], in_val[kSlots];       // per shard, on devices[0]: where its lists arrive
    DevMem out_kp[kSlots], out_desc[kSlots];                  // on devices[0]: gathered lists, global image order
    hipStream_t recv_stream = nullptr, copy_stream = nullptr;   // on devices[0]
    hipStream_t user_stream = nullptr;                          // on devices[0]: the caller's copies of a collected batch (never the null stream)
    void* user_stage[2] = {nullptr, nullptr};                   // page-locked, kUserStage bytes each: ordinary caller memory is reached through them
    hipEvent_t user_ev[2] = {nullptr, nullptr};
    // the batch the result accessors read (the last one collected)
    int cur = -1;
    double compute_ms = 0, gather_ms = 0, exposed_ms = 0;
    long long gather_bytes = 0;
};

namespace {

bool distinct(const std::vector<int>& d) {
    for (size_t i = 0; i < d.size(); ++i)
        for (size_t j = i + 1; j < d.size(); ++j)
            if (d[i] == d[j]) return false;
    return true;
}

// RCCL communicators, made on first use (the options are known by then).  Holding g->m.
void init_transport(sift_hip_group* g) {
    if (g->comm_tried) return;
    g->comm_tried = true;
    g->use_rccl = false;
    const bool loop = g->S == 1 && g->gather_loopback;
    if (g->gather_transport == 0) { g->rccl_note = "copies (option gather_transport = 0)"; return; }
    if (g->S == 1 && !loop) { g->rccl_note = "one shard: nothing to gather"; return; }
    if (!distinct(g->devices)) { g->rccl_note = "copies: RCCL takes one rank per GPU and the group lists a device twice"; return; }
    Rccl& r = rccl();
    if (!r.ok) { g->rccl_note = "copies: librccl.so.1 not found"; return; }
    g->comm.assign((size_t)g->S, nullptr);
    const ncclResult_t e = r.CommInitAll(g->comm.data(), g->S, g->devices.data());
    if (e != ncclSuccess) {
        g->rccl_note = std::string("copies: ncclCommInitAll failed: ") + r.GetErrorString(e);
        g->comm.clear();
        return;
    }
    g->use_rccl = true;
    g->rccl_note = "RCCL point-to-point (ncclSend / ncclRecv), one communicator per GPU";
}

bool remote(const sift_hip_group* g, int s) { return g->devices[(size_t)s] != g->devices[0]; }
bool wants_pack(const sift_hip_group* g, int s) {
    return g->gather_wire == 2 || (g->gather_wire == 1 && (remote(g, s) || (g->S == 1 && g->gather_loopback)));
}

// ---- shard thread ----------------------------------------------------------------------------------------------------
void shard_main(sift_hip_group* g, int s) {
    const int dev = g->devices[(size_t)s];
    (void)hipSetDevice(dev);
    for (;;) {
        long long seq;
        {
            std::unique_lock<std::mutex> lk(g->m);
            g->cv.wait(lk, [&] { return g->stop || g->shard_next[(size_t)s] < g->submitted; });
            if (g->stop) return;
            seq = g->shard_next[(size_t)s];
        }
        const int slot = (int)(seq % kSlots);
        Batch& B = g->batch[slot];
        ShardBatch& R = B.shard[(size_t)s];
        const int cnt = B.count[(size_t)s];
        bool wait_for_transfer = false;
        if (cnt > 0) {
            try {
                char e[512] = "";
                const double t0 = now_ms();
                const size_t frame = (size_t)B.w * (size_t)B.h;
                R.rc = sift_hip_calculate_batch(g->ctx[(size_t)s], B.imgs + (size_t)B.first[(size_t)s] * frame, cnt, B.w, B.h, &B.params, e, sizeof(e));
                R.msg = e;
                R.ran = sift_hip_result_images(g->ctx[(size_t)s]) == cnt;
                if (R.ran) {
                    R.status.assign((size_t)cnt, 0);
                    R.counts.assign((size_t)cnt, 0);
                    (void)sift_hip_result_status(g->ctx[(size_t)s], R.status.data(), cnt);
                    (void)sift_hip_result_counts(g->ctx[(size_t)s], R.counts.data(), cnt);
                    R.total = sift_hip_result_total(g->ctx[(size_t)s]);
                }
                // lists that cross to another GPU are packed here, on the shard's own GPU and thread
                if (R.ran && R.total > 0 && wants_pack(g, s)) {
                    int64_t nnz = 0;
                    int lossless = 0;
                    if (sift_hip_result_sparse_size(g->ctx[(size_t)s], &nnz, &lossless) == SIFT_HIP_OK && lossless &&
                        g->pack_rec[slot][(size_t)s].fit(dev, R.total * 34) && g->pack_val[slot][(size_t)s].fit(dev, std::max<long long>(nnz, 1) * 4) &&
                        sift_hip_result_sparse_pack(g->ctx[(size_t)s], g->pack_rec[slot][(size_t)s].p, g->pack_val[slot][(size_t)s].p) == SIFT_HIP_OK) {
                        R.nnz = nnz;
                        R.packed = true;
                    }
                }
                R.compute_ms = now_ms() - t0;
                // ---- the transfer, queued by the shard itself on its own stream as soon as its lists exist -------------------
                if (R.ran && R.total > 0) {
                    (void)hipSetDevice(dev);
                    const bool loop = g->S == 1 && g->gather_loopback && g->use_rccl;
                    const void* src_a = nullptr;
                    const void* src_b = nullptr;
                    size_t na = 0, nb = 0;
                    if (R.packed) {
                        src_a = g->pack_rec[slot][(size_t)s].p; na = (size_t)R.total * 34;
                        src_b = g->pack_val[slot][(size_t)s].p; nb = (size_t)R.nnz * 4;
                    } else {
                        (void)sift_hip_result_device(g->ctx[(size_t)s], &src_a, &src_b);
                        na = (size_t)R.total * sizeof(sift_hip_keypoint); nb = (size_t)R.total * 128 * sizeof(float);
                    }
                    if (loop) {
                        // one shard, one GPU: the gather thread sends and receives (a communicator serves one thread at a time)
                    } else if (g->use_rccl && s > 0) {
                        // RCCL: this GPU's side of the exchange; the gather thread posts the matching receives in one group
                        Rccl& r = rccl();
                        ncclResult_t e1 = r.GroupStart();
                        if (e1 == ncclSuccess) e1 = r.Send(src_a, na, ncclUint8, 0, g->comm[(size_t)s], g->send_stream[(size_t)s]);
                        if (e1 == ncclSuccess && nb) e1 = r.Send(src_b, nb, ncclUint8, 0, g->comm[(size_t)s], g->send_stream[(size_t)s]);
                        const ncclResult_t e2 = r.GroupEnd();
                        if (e1 != ncclSuccess || e2 != ncclSuccess) {
                            // nothing was queued: no receive may be posted for this shard (it would never complete)
                            R.rc = SIFT_HIP_EHIP;
                            R.msg = std::string("sift_hip_group: ncclSend failed: ") + r.GetErrorString(e1 != ncclSuccess ? e1 : e2);
                        } else {
                            R.sent = true;
                        }
                    } else {
                        // copies: straight into this shard's arrival area on devices[0], over this GPU's own link (a shard on
                        // devices[0] itself moves its lists aside the same way: its context is then free for the next batch)
                        DevMem& ia = g->in_rec[slot][(size_t)s];
                        DevMem& ib = g->in_val[slot][(size_t)s];
                        const bool ok = ia.fit(g->devices[0], (long long)na) && ib.fit(g->devices[0], (long long)std::max<size_t>(nb, 4));
                        (void)hipSetDevice(dev);
                        hipError_t h1 = hipSuccess, h2 = hipSuccess;
                        if (ok && g->copy_kernels && g->peer_ok[(size_t)s] && ((reinterpret_cast<uintptr_t>(src_a) | reinterpret_cast<uintptr_t>(src_b)) & 3u) == 0 && na % 4 == 0 && nb % 4 == 0) {
                            // a kernel of the library, run by THIS GPU: it writes the arrival area in place (its own memory, or the
                            // first GPU's through peer access over the link)
                            if (sift_hip_internal_copy(g->send_stream[(size_t)s], src_a, ia.p, na)) h1 = hipErrorUnknown;
                            if (nb && sift_hip_internal_copy(g->send_stream[(size_t)s], src_b, ib.p, nb)) h2 = hipErrorUnknown;
                        } else if (ok && remote(g, s)) {
                            h1 = hipMemcpyPeerAsync(ia.p, g->devices[0], src_a, dev, na, g->send_stream[(size_t)s]);
                            if (nb) h2 = hipMemcpyPeerAsync(ib.p, g->devices[0], src_b, dev, nb, g->send_stream[(size_t)s]);
                        } else if (ok) {
                            h1 = hipMemcpyAsync(ia.p, src_a, na, hipMemcpyDeviceToDevice, g->send_stream[(size_t)s]);
                            if (nb) h2 = hipMemcpyAsync(ib.p, src_b, nb, hipMemcpyDeviceToDevice, g->send_stream[(size_t)s]);
                        }
                        if (!ok || h1 != hipSuccess || h2 != hipSuccess) {
                            R.rc = SIFT_HIP_EHIP;
                            R.msg = ok ? "sift_hip_group: device-to-device copy failed" : "sift_hip_group: out of device memory for the arriving lists";
                            (void)hipGetLastError();
                        } else {
                            R.sent = true;
                        }
                    }
                    (void)hipEventRecord(g->sent_ev[slot][(size_t)s], g->send_stream[(size_t)s]);
                    wait_for_transfer = !R.packed && R.sent;
                }
            } catch (const std::exception& ex) {
                R.rc = SIFT_HIP_EHIP;
                R.msg = std::string("sift_hip_group: ") + ex.what();
            }
        }
        {
            std::lock_guard<std::mutex> lk(g->m);
            if (++B.reported == g->S) B.t_computed = now_ms();
        }
        g->cv.notify_all();
        // Lists that leave from the context's own arrays (not packed) must be gone before the context runs its next batch.  Only
        // now, after reporting: an RCCL send completes when the gather thread has posted its receive, which waits for the reports.
        if (wait_for_transfer) (void)hipStreamSynchronize(g->send_stream[(size_t)s]);
        {
            std::lock_guard<std::mutex> lk(g->m);
            g->shard_next[(size_t)s] = seq + 1;
        }
        g->cv.notify_all();
    }
}

// ---- gather thread: keypoint lists only, device to device, global image order ------------------------------------------
void gather_batch(sift_hip_group* g, Batch& B, int slot) {
    const int S = g->S;
    B.status.assign((size_t)B.n, 0);
    B.counts.assign((size_t)B.n, 0);
    B.rc = SIFT_HIP_OK;
    B.msg.clear();
    B.total = 0;
    B.wire_bytes = 0;
    std::vector<long long> off((size_t)S, 0);
    for (int s = 0; s < S; ++s) {
        const ShardBatch& R = B.shard[(size_t)s];
        const int cnt = B.count[(size_t)s];
        if (cnt == 0) continue;
        if (R.rc != SIFT_HIP_OK && B.rc == SIFT_HIP_OK) { B.rc = R.rc; B.msg = R.msg; }
        if (!R.ran) {   // nothing ran on this shard: the batch has no result
            if (B.rc == SIFT_HIP_OK) { B.rc = SIFT_HIP_EHIP; B.msg = "sift_hip_group_calculate: a shard returned no results"; }
            B.total = -1;
            break;
        }
        std::copy(R.status.begin(), R.status.end(), B.status.begin() + B.first[(size_t)s]);
        std::copy(R.counts.begin(), R.counts.end(), B.counts.begin() + B.first[(size_t)s]);
        off[(size_t)s] = B.total;
        B.total += R.total;
    }
    auto fail = [&](const std::string& m) {
        B.rc = SIFT_HIP_EHIP;
        B.msg = m;
        B.total = -1;
    };
    // Receives must be posted whatever happened above: a shard that has queued ncclSend waits for them.
    const int dev0 = g->devices[0];
    (void)hipSetDevice(dev0);
    Rccl& r = rccl();
    const bool loop = S == 1 && g->gather_loopback && g->use_rccl;
    if (g->use_rccl) {
        bool any = false, ok = true;
        for (int s = loop ? 0 : 1; s < S; ++s) {
            const ShardBatch& R = B.shard[(size_t)s];
            if (!(R.sent || (loop && R.ran && R.total > 0))) continue;
            const long long na = R.packed ? R.total * 34 : R.total * (long long)sizeof(sift_hip_keypoint);
            const long long nb = R.packed ? R.nnz * 4 : R.total * 128 * (long long)sizeof(float);
            ok = g->in_rec[slot][(size_t)s].fit(dev0, na) && g->in_val[slot][(size_t)s].fit(dev0, std::max<long long>(nb, 4)) && ok;
            any = true;
        }
        if (any && ok) {
            // ONE group: every link carries its shard's lists at the same time
            ncclResult_t e1 = r.GroupStart();
            for (int s = loop ? 0 : 1; s < S && e1 == ncclSuccess; ++s) {
                const ShardBatch& R = B.shard[(size_t)s];
                if (!(R.sent || (loop && R.ran && R.total > 0))) continue;
                const size_t na = R.packed ? (size_t)R.total * 34 : (size_t)R.total * sizeof(sift_hip_keypoint);
                const size_t nb = R.packed ? (size_t)R.nnz * 4 : (size_t)R.total * 128 * sizeof(float);
                if (loop) {   // one shard, one GPU: the sends are this thread's too (a communicator serves one thread at a time)
                    const void* src_a = g->pack_rec[slot][0].p;
                    const void* src_b = g->pack_val[slot][0].p;
                    if (!R.packed) (void)sift_hip_result_device(g->ctx[0], &src_a, &src_b);
                    e1 = r.Send(src_a, na, ncclUint8, 0, g->comm[0], g->recv_stream);
                    if (e1 == ncclSuccess && nb) e1 = r.Send(src_b, nb, ncclUint8, 0, g->comm[0], g->recv_stream);
                }
                if (e1 == ncclSuccess) e1 = r.Recv(g->in_rec[slot][(size_t)s].p, na, ncclUint8, s, g->comm[0], g->recv_stream);
                if (e1 == ncclSuccess && nb) e1 = r.Recv(g->in_val[slot][(size_t)s].p, nb, ncclUint8, s, g->comm[0], g->recv_stream);
                B.wire_bytes += (long long)(na + nb);
            }
            const ncclResult_t e2 = r.GroupEnd();
            if (e1 != ncclSuccess || e2 != ncclSuccess) fail(std::string("sift_hip_group: ncclRecv failed: ") + r.GetErrorString(e1 != ncclSuccess ? e1 : e2));
            else if (hipStreamSynchronize(g->recv_stream) != hipSuccess) fail("sift_hip_group: the RCCL gather failed");
        } else if (any) {
            fail("sift_hip_group: out of device memory for the arriving lists");
        }
    } else {
        for (int s = 0; s < S; ++s) {   // copies: wait for every shard's own stream
            const ShardBatch& R = B.shard[(size_t)s];
            if (!R.sent) continue;
            if (hipEventSynchronize(g->sent_ev[slot][(size_t)s]) != hipSuccess) fail("sift_hip_group: a device-to-device copy failed");
            if (remote(g, s)) B.wire_bytes += R.packed ? R.total * 34 + R.nnz * 4 : R.total * (20 + 512);
        }
    }
    if (B.total < 0) return;
    if (B.total > 0 && (!g->out_kp[slot].fit(dev0, B.total * (long long)sizeof(sift_hip_keypoint)) ||
                        !g->out_desc[slot].fit(dev0, B.total * 128 * (long long)sizeof(float)))) {
        fail("sift_hip_group_calculate: out of device memory for the gathered lists");
        return;
    }
    (void)hipSetDevice(dev0);
    for (int s = 0; s < S; ++s) {
        const ShardBatch& R = B.shard[(size_t)s];
        if (R.total <= 0) continue;
        char* dk = static_cast<char*>(g->out_kp[slot].p) + (size_t)off[(size_t)s] * sizeof(sift_hip_keypoint);
        char* dd = static_cast<char*>(g->out_desc[slot].p) + (size_t)off[(size_t)s] * 128 * sizeof(float);
        const bool arrived = R.sent || (loop && s == 0);
        if (R.packed) {
            const void* rec = arrived ? g->in_rec[slot][(size_t)s].p : g->pack_rec[slot][(size_t)s].p;
            const void* val = arrived ? g->in_val[slot][(size_t)s].p : g->pack_val[slot][(size_t)s].p;
            if (sift_hip_sparse_unpack(g->unpack_ctx, rec, val, R.total, dk, dd) != SIFT_HIP_OK) { fail("sift_hip_group_calculate: unpacking the arriving lists failed"); return; }
        } else {
            const void *kp = nullptr, *desc = nullptr;
            if (arrived) { kp = g->in_rec[slot][(size_t)s].p; desc = g->in_val[slot][(size_t)s].p; }
            else if (sift_hip_result_device(g->ctx[(size_t)s], &kp, &desc) != SIFT_HIP_OK) { fail("sift_hip_group: no device results"); return; }
            // (lists still in the shard's own buffers: loopback mode only, which runs one batch at a time)
            const bool al = ((reinterpret_cast<uintptr_t>(kp) | reinterpret_cast<uintptr_t>(desc) | reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dd)) & 3u) == 0;
            if (g->copy_kernels && al) {
                if (sift_hip_internal_copy(g->copy_stream, kp, dk, (size_t)R.total * sizeof(sift_hip_keypoint)) ||
                    sift_hip_internal_copy(g->copy_stream, desc, dd, (size_t)R.total * 128 * sizeof(float))) {
                    fail("sift_hip_group: device-to-device copy failed");
                    return;
                }
            } else if (hipMemcpyAsync(dk, kp, (size_t)R.total * sizeof(sift_hip_keypoint), hipMemcpyDeviceToDevice, g->copy_stream) != hipSuccess ||
                hipMemcpyAsync(dd, desc, (size_t)R.total * 128 * sizeof(float), hipMemcpyDeviceToDevice, g->copy_stream) != hipSuccess) {
                fail("sift_hip_group: device-to-device copy failed");
                return;
            }
        }
    }
    if (hipStreamSynchronize(g->copy_stream) != hipSuccess) fail("sift_hip_group_calculate: gather failed");
}

void gather_main(sift_hip_group* g) {
    (void)hipSetDevice(g->devices[0]);
    for (;;) {
        long long seq;
        {
            std::unique_lock<std::mutex> lk(g->m);
            g->cv.wait(lk, [&] { return g->stop || (g->gather_next < g->submitted && g->batch[g->gather_next % kSlots].reported == g->S); });
            if (g->stop) return;
            seq = g->gather_next;
        }
        Batch& B = g->batch[seq % kSlots];
        try {
            gather_batch(g, B, (int)(seq % kSlots));
        } catch (const std::exception& ex) {
            B.rc = SIFT_HIP_EHIP;
            B.msg = std::string("sift_hip_group: ") + ex.what();
            B.total = -1;
        }
        {
            std::lock_guard<std::mutex> lk(g->m);
            B.t_gathered = now_ms();
            B.gathered = true;
            g->gather_next = seq + 1;
        }
        g->cv.notify_all();
    }
}

}  // namespace

extern "C" {

int sift_hip_group_create(const int* devices, int n_devices, sift_hip_group** out, char* err, int errlen) {
    if (!devices || n_devices <= 0 || !out) return SIFT_HIP_EINVAL;
    *out = nullptr;
    auto* g = new sift_hip_group();
    g->devices.assign(devices, devices + n_devices);
    g->S = n_devices;
    auto bail = [&](int rc, const char* m) {
        if (m) set_err(err, errlen, m);
        for (auto* p : g->ctx) sift_hip_destroy(p);
        if (g->unpack_ctx) sift_hip_destroy(g->unpack_ctx);
        delete g;
        return rc;
    };
    for (int s = 0; s < n_devices; ++s) {
        sift_hip_ctx* c = nullptr;
        const int rc = sift_hip_create(devices[s], &c, err, errlen);
        if (rc != SIFT_HIP_OK) return bail(rc, nullptr);
        g->ctx.push_back(c);
    }
    {
        const int rc = sift_hip_create(devices[0], &g->unpack_ctx, err, errlen);
        if (rc != SIFT_HIP_OK) return bail(rc, nullptr);
    }
    // direct peer copies into shard 0's device (already-enabled and same-device answers are fine)
    // (recorded per shard: only a GPU that CAN write the first GPU's memory in place runs the library's copy kernel against it -
    // anywhere else that would be a memory fault that ends the process; the runtime's peer copy stages through the host instead)
    g->peer_ok.assign((size_t)n_devices, 1);
    for (int s = 1; s < n_devices; ++s)
        if (devices[s] != devices[0]) {
            int can = 0;
            bool ok = false;
            if (hipDeviceCanAccessPeer(&can, devices[s], devices[0]) == hipSuccess && can) {
                ApiGuard api(devices[s]);
                (void)hipSetDevice(devices[s]);
                const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
                ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            }
            (void)hipGetLastError();
            g->peer_ok[(size_t)s] = ok ? 1 : 0;
        }
    g->send_stream.assign((size_t)n_devices, nullptr);
    for (int b = 0; b < kSlots; ++b) {
        g->sent_ev[b].assign((size_t)n_devices, nullptr);
        g->pack_rec[b].resize((size_t)n_devices); g->pack_val[b].resize((size_t)n_devices);
        g->in_rec[b].resize((size_t)n_devices); g->in_val[b].resize((size_t)n_devices);
    }
    for (int s = 0; s < n_devices; ++s) {
        ApiGuard api(devices[s]);
        if (hipSetDevice(devices[s]) != hipSuccess || hipStreamCreateWithFlags(&g->send_stream[(size_t)s], hipStreamNonBlocking) != hipSuccess)
            return bail(SIFT_HIP_EHIP, "sift_hip_group_create: cannot create a shard's transfer stream");
        for (int b = 0; b < kSlots; ++b)
            if (hipEventCreateWithFlags(&g->sent_ev[b][(size_t)s], hipEventDisableTiming) != hipSuccess)
                return bail(SIFT_HIP_EHIP, "sift_hip_group_create: cannot create an event");
    }
    ApiGuard api(devices[0]);
    (void)hipSetDevice(devices[0]);
    if (hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&g->user_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&g->recv_stream, hipStreamNonBlocking) != hipSuccess)
        return bail(SIFT_HIP_EHIP, "sift_hip_group_create: cannot create the gather streams");
    g->shard_next.assign((size_t)n_devices, 0);
    try {
        for (int s = 0; s < n_devices; ++s) g->workers.emplace_back(shard_main, g, s);
        g->gatherer = std::thread(gather_main, g);
    } catch (const std::exception& e) {
        {
            std::lock_guard<std::mutex> lk(g->m);
            g->stop = true;
        }
        g->cv.notify_all();
        for (auto& t : g->workers) if (t.joinable()) t.join();
        return bail(SIFT_HIP_EHIP, e.what());
    }
    *out = g;
    return SIFT_HIP_OK;
}

void sift_hip_group_destroy(sift_hip_group* g) {
    if (!g) return;
    {
        std::unique_lock<std::mutex> lk(g->m);
        // batches still in flight run to their end first (a shard may be inside an RCCL exchange another thread has to answer)
        g->cv.wait(lk, [&] { return g->gather_next == g->submitted; });
        g->stop = true;
    }
    g->cv.notify_all();
    for (auto& t : g->workers) if (t.joinable()) t.join();
    if (g->gatherer.joinable()) g->gatherer.join();
    if (!g->comm.empty()) {
        Rccl& r = rccl();
        for (auto c : g->comm) if (c) (void)r.CommDestroy(c);
    }
    for (int s = 0; s < g->S; ++s) {
        ApiGuard api(g->devices[(size_t)s]);
        (void)hipSetDevice(g->devices[(size_t)s]);
        if (g->send_stream[(size_t)s]) { (void)hipStreamSynchronize(g->send_stream[(size_t)s]); (void)hipStreamDestroy(g->send_stream[(size_t)s]); }
        for (int b = 0; b < kSlots; ++b) {
            if (g->sent_ev[b][(size_t)s]) (void)hipEventDestroy(g->sent_ev[b][(size_t)s]);
            g->pack_rec[b][(size_t)s].release(g->devices[(size_t)s]);
            g->pack_val[b][(size_t)s].release(g->devices[(size_t)s]);
            g->in_rec[b][(size_t)s].release(g->devices[0]);
            g->in_val[b][(size_t)s].release(g->devices[0]);
        }
    }
    ApiGuard api0(g->devices[0]);
    (void)hipSetDevice(g->devices[0]);
    for (int b = 0; b < kSlots; ++b) { g->out_kp[b].release(g->devices[0]); g->out_desc[b].release(g->devices[0]); }
    if (g->copy_stream) { (void)hipStreamSynchronize(g->copy_stream); (void)hipStreamDestroy(g->copy_stream); }
    if (g->recv_stream) { (void)hipStreamSynchronize(g->recv_stream); (void)hipStreamDestroy(g->recv_stream); }
    if (g->user_stream) { (void)hipStreamSynchronize(g->user_stream); (void)hipStreamDestroy(g->user_stream); }
    for (int b = 0; b < 2; ++b) {
        if (g->user_ev[b]) (void)hipEventDestroy(g->user_ev[b]);
        if (g->user_stage[b]) sift_hip_host_free(g->user_stage[b]);
    }
    for (auto* c : g->ctx) sift_hip_destroy(c);
    if (g->unpack_ctx) sift_hip_destroy(g->unpack_ctx);
    delete g;
}

int sift_hip_group_shards(sift_hip_group* g) { return g ? g->S : -1; }

int sift_hip_group_set_option(sift_hip_group* g, const char* name, int value) {
    if (!g || !name) return SIFT_HIP_EINVAL;
    {
        std::lock_guard<std::mutex> lk(g->m);
        if (g->collected != g->submitted) return SIFT_HIP_EINVAL;   // not while a batch is in flight
        if (!std::strcmp(name, "gather_wire")) {
            if (value < 0 || value > 2) return SIFT_HIP_EINVAL;
            g->gather_wire = value;
            return SIFT_HIP_OK;
        }
        if (!std::strcmp(name, "copy_kernels")) { g->copy_kernels = value != 0; return SIFT_HIP_OK; }
        if (!std::strcmp(name, "gather_transport")) {   // before the first batch: the communicators are made then
            if (value < 0 || value > 2 || g->comm_tried) return SIFT_HIP_EINVAL;
            g->gather_transport = value;
            return SIFT_HIP_OK;
        }
        if (!std::strcmp(name, "gather_loopback")) {
            if (value < 0 || value > 1 || g->comm_tried) return SIFT_HIP_EINVAL;
            g->gather_loopback = value;
            return SIFT_HIP_OK;
        }
    }
    int rc = SIFT_HIP_OK;
    for (auto* c : g->ctx) rc = std::max(rc, sift_hip_set_option(c, name, value));
    return rc;
}

int sift_hip_group_transport(sift_hip_group* g, char* text, int textlen) {
    if (!g) return -1;
    std::lock_guard<std::mutex> lk(g->m);
    init_transport(g);
    set_err(text, textlen, g->rccl_note);
    return g->use_rccl ? 1 : 0;
}

int sift_hip_group_submit(sift_hip_group* g, const float* host_imgs, int n, int w, int h, const sift_hip_params* params, char* err, int errlen) {
    if (!g || !host_imgs || !params || n <= 0 || w <= 0 || h <= 0) return SIFT_HIP_EINVAL;
    try {
        std::unique_lock<std::mutex> lk(g->m);
        if (g->submitted - g->collected >= kSlots) {
            set_err(err, errlen, "sift_hip_group_submit: two batches are in flight already: collect one first");
            return SIFT_HIP_EINVAL;
        }
        init_transport(g);
        if (g->gather_transport == 2 && !g->use_rccl && (g->S > 1 || g->gather_loopback)) {
            set_err(err, errlen, "sift_hip_group_submit: option gather_transport = 2 asks for RCCL: " + g->rccl_note);
            return SIFT_HIP_EHIP;
        }
        if (g->S == 1 && g->gather_loopback)   // test mode: the gather thread reads the shard's own buffers, one batch at a time
            g->cv.wait(lk, [&] { return g->gather_next == g->submitted; });
        const long long seq = g->submitted;
        Batch& B = g->batch[seq % kSlots];
        B = Batch();
        B.seq = seq;
        B.n = n; B.w = w; B.h = h; B.imgs = host_imgs; B.params = *params;
        const int S = g->S;
        B.first.assign((size_t)S, 0);
        B.count.assign((size_t)S, 0);
        B.shard.assign((size_t)S, ShardBatch());
        const int per = (n + S - 1) / S;   // contiguous blocks: shard s holds images s*per .. (256 -> 32 each on 8 GPUs)
        for (int s = 0; s < S; ++s) {
            B.first[(size_t)s] = std::min(n, s * per);
            B.count[(size_t)s] = std::min(n, (s + 1) * per) - B.first[(size_t)s];
        }
        B.t_submit = now_ms();
        if (g->cur == (int)(seq % kSlots)) g->cur = -1;   // the results read so far lived in this slot
        if (g->gather_wire)
            for (auto* c : g->ctx) (void)sift_hip_set_option(c, "wire_count", 1);   // the descriptor kernels count what the wire will carry
        g->submitted = seq + 1;
    } catch (const std::exception& e) {
        set_err(err, errlen, std::string("sift_hip_group_submit: ") + e.what());
        return SIFT_HIP_EHIP;
    }
    g->cv.notify_all();
    return SIFT_HIP_OK;
}

int sift_hip_group_collect(sift_hip_group* g, char* err, int errlen) {
    if (!g) return SIFT_HIP_EINVAL;
    std::unique_lock<std::mutex> lk(g->m);
    if (g->collected == g->submitted) {
        set_err(err, errlen, "sift_hip_group_collect: no batch in flight");
        return SIFT_HIP_EINVAL;
    }
    const long long seq = g->collected;
    const double t_wait = now_ms();
    g->cv.wait(lk, [&] { return g->gather_next > seq; });
    Batch& B = g->batch[seq % kSlots];
    g->collected = seq + 1;
    g->cur = B.total >= 0 ? (int)(seq % kSlots) : -1;
    double cms = 0;
    for (const auto& R : B.shard) cms = std::max(cms, R.compute_ms);
    g->compute_ms = cms;
    g->gather_ms = B.t_gathered - B.t_computed;                 // last shard done -> lists in place on devices[0]
    g->exposed_ms = std::max(0.0, B.t_gathered - std::max(t_wait, B.t_computed));   // the part of it this call had to wait for
    g->gather_bytes = B.wire_bytes;
    if (B.rc != SIFT_HIP_OK) set_err(err, errlen, B.msg);
    return B.rc;
}

int sift_hip_group_calculate(sift_hip_group* g, const float* host_imgs, int n, int w, int h, const sift_hip_params* params,
                             char* err, int errlen) {
    if (!g) return SIFT_HIP_EINVAL;
    {
        std::lock_guard<std::mutex> lk(g->m);
        if (g->collected != g->submitted) {
            set_err(err, errlen, "sift_hip_group_calculate: batches submitted earlier have not been collected");
            return SIFT_HIP_EINVAL;
        }
    }
    const int rc = sift_hip_group_submit(g, host_imgs, n, w, h, params, err, errlen);
    if (rc != SIFT_HIP_OK) {
        std::lock_guard<std::mutex> lk(g->m);
        g->cur = -1;
        return rc;
    }
    return sift_hip_group_collect(g, err, errlen);
}

int sift_hip_group_result_images(sift_hip_group* g) { return (g && g->cur >= 0) ? g->batch[g->cur].n : -1; }
int64_t sift_hip_group_result_total(sift_hip_group* g) { return (g && g->cur >= 0) ? g->batch[g->cur].total : -1; }
int sift_hip_group_result_status(sift_hip_group* g, int32_t* status, int cap) {
    if (!g || g->cur < 0 || !status || cap < g->batch[g->cur].n) return SIFT_HIP_EINVAL;
    std::copy(g->batch[g->cur].status.begin(), g->batch[g->cur].status.end(), status);
    return SIFT_HIP_OK;
}
int sift_hip_group_result_counts(sift_hip_group* g, int32_t* counts, int cap) {
    if (!g || g->cur < 0 || !counts || cap < g->batch[g->cur].n) return SIFT_HIP_EINVAL;
    std::copy(g->batch[g->cur].counts.begin(), g->batch[g->cur].counts.end(), counts);
    return SIFT_HIP_OK;
}
int sift_hip_group_result_device(sift_hip_group* g, const void** dev_keypoints, const void** dev_descriptors) {
    if (!g || g->cur < 0) return SIFT_HIP_EINVAL;
    if (dev_keypoints) *dev_keypoints = g->out_kp[g->cur].p;
    if (dev_descriptors) *dev_descriptors = g->out_desc[g->cur].p;
    return SIFT_HIP_OK;
}
int sift_hip_group_result_copy(sift_hip_group* g, sift_hip_keypoint* kp, float* desc) {
    if (!g || g->cur < 0) return SIFT_HIP_EINVAL;
    const long long total = g->batch[g->cur].total;
    if (total <= 0) return SIFT_HIP_OK;
    if (hipSetDevice(g->devices[0]) != hipSuccess) return SIFT_HIP_EHIP;
    // A stream of its own (not the gather thread's: the next batch's gather may be running; not the null stream).  Ordinary
    // (pageable) caller memory is NOT handed to the runtime: its copy would page-lock the caller's pages for the transfer and
    // release them afterwards, and with other threads launching kernels at that moment the launches crashed inside the runtime
    // (3 - 8 % of the runs of examples/sift_multi_gpu.cpp, tools/example_loop.sh).  Such memory is reached through two
    // page-locked staging buffers of the group, chunk k+1 on its way while chunk k is copied out by the host.
    auto pull = [&](void* dst, const void* src, size_t bytes) -> bool {
        hipPointerAttribute_t a;
        std::memset(&a, 0, sizeof(a));
        const bool direct = hipPointerGetAttributes(&a, dst) == hipSuccess && (a.type == hipMemoryTypeHost || a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged);
        (void)hipGetLastError();
        if (direct) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, g->user_stream) == hipSuccess && hipStreamSynchronize(g->user_stream) == hipSuccess;
        constexpr size_t kUserStage = 16u << 20;
        for (int b = 0; b < 2; ++b) {
            if (!g->user_stage[b]) g->user_stage[b] = sift_hip_host_alloc(kUserStage);
            if (!g->user_stage[b]) return false;
            if (!g->user_ev[b]) {
                ApiGuard api;
                if (hipEventCreateWithFlags(&g->user_ev[b], hipEventDisableTiming) != hipSuccess) return false;
            }
        }
        const size_t chunks = (bytes + kUserStage - 1) / kUserStage;
        for (size_t k = 0; k <= chunks; ++k) {
            if (k < chunks) {
                const size_t n = std::min(kUserStage, bytes - k * kUserStage);
                if (hipMemcpyAsync(g->user_stage[k & 1], static_cast<const char*>(src) + k * kUserStage, n, hipMemcpyDeviceToHost, g->user_stream) != hipSuccess ||
                    hipEventRecord(g->user_ev[k & 1], g->user_stream) != hipSuccess)
                    return false;
            }
            if (k >= 1) {
                const size_t o = (k - 1) * kUserStage, n = std::min(kUserStage, bytes - o);
                if (hipEventSynchronize(g->user_ev[(k - 1) & 1]) != hipSuccess) return false;
                std::memcpy(static_cast<char*>(dst) + o, g->user_stage[(k - 1) & 1], n);
            }
        }
        return true;
    };
    if (kp && !pull(kp, g->out_kp[g->cur].p, (size_t)total * sizeof(sift_hip_keypoint))) return SIFT_HIP_EHIP;
    if (desc && !pull(desc, g->out_desc[g->cur].p, (size_t)total * 128 * sizeof(float))) return SIFT_HIP_EHIP;
    return SIFT_HIP_OK;
}
int sift_hip_group_timing(sift_hip_group* g, double* compute_ms, double* gather_ms, int64_t* gather_bytes) {
    if (!g || g->collected == 0) return SIFT_HIP_EINVAL;
    if (compute_ms) *compute_ms = g->compute_ms;
    if (gather_ms) *gather_ms = g->gather_ms;
    if (gather_bytes) *gather_bytes = g->gather_bytes;
    return SIFT_HIP_OK;
}
/* Time the process's host threads have spent waiting for a launch lock (launch_guard.h: one per device) since it started. */
int sift_hip_lock_wait_ms(double* ms) {
    if (!ms) return SIFT_HIP_EINVAL;
    *ms = sift_hip::launch_lock_wait_ms();
    return SIFT_HIP_OK;
}
int sift_hip_group_gather_exposed(sift_hip_group* g, double* exposed_ms) {
    if (!g || g->collected == 0 || !exposed_ms) return SIFT_HIP_EINVAL;
    *exposed_ms = g->exposed_ms;
    return SIFT_HIP_OK;
}

}  // extern "C"
