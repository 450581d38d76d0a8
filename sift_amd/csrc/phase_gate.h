// Phase gate: orders the phases of consecutive batches that run on DIFFERENT contexts of one GPU.
//
// One batch of Sift::calculate (sift.cpp:19-57) has four stretches on the device:
//     P  pyramid (bandwidth-bound, fills the chip, the launches the roofline figure is measured on)
//     E  extrema + edge filter, gradient maps (fill the chip)
//     C  cleanup -> orientation -> cleanup (one workgroup per image: an eighth of the chip for ~0.6 ms)
//     D  descriptors (fill the chip)
// With two contexts taking batches alternately, the gate makes the device run (schedule 1, the default since round 3)
//     ... | P(g+1) || C(g) | E(g+1) || D(g) | P(g+2) || C(g+1) | E(g+2) || D(g+1) | ...
// i.e. the cleanup chain, which cannot fill the chip, runs under the next batch's pyramid, and the descriptors (issue-bound)
// under the next batch's extrema / gradient pass; no two pyramids, and no two descriptor stages, ever share the chip.
// Schedule 0 (rounds 1 - 2; option "gate_schedule" = 0) keeps the pyramids alone on the chip:
//     ... | P(g+1) | C(g) || E(g+1), then D(g) | P(g+2) | C(g+1) || E(g+2), then D(g+1) | ...
// The blur launches then run at their stand-alone rate (0.50 of the HBM peak against 0.43 beside the chain), but the chain
// starves behind the extrema pass's persistent workgroups and the descriptors wait for it: 2.98 against 2.82 - 2.95 ms per step
// (round 3; in round 2, before the reductions evaluated kept pixels only and the fills left the streams, schedule 1 was the
// slower one).  Everything is expressed
// with events between the contexts' streams; the host side only makes sure an event has been RECORDED before
// another stream is told to wait for it (a wait on an unrecorded event is no wait at all).
//
// Rules of schedule 0, g = ticket of a batch in submission order:
//     P(g)  starts after E(g-1), or after D(g-1) if batch g-1 had already entered C when g was announced;
//           and after D(g-2).
//     C(g)  starts after P(g+1) if batch g+1 has been announced by the time the device finishes E(g), else then.
// (So three contexts are needed for two batches to overlap all the time: batch g+1 must be submitted while g-1 is
// still running, i.e. before its owner has seen g-2's results.)
// A batch that fails half way releases everything it owes (finish()), so a partner never waits for a dead batch.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <mutex>

namespace sift_hip {

class PhaseGate {
public:
    enum Phase { kP = 0, kE = 1, kD = 2, kC = 3 };
    // 0: the schedule above.  1 (option "gate_schedule"): the cleanup chain C(g) runs under the next pyramid P(g+1) and the
    // descriptors D(g) under the next extrema / gradient pass E(g+1):  ... | P(g+1) || C(g) | E(g+1) || D(g) | P(g+2) || C(g+1) | ...
    void set_schedule(int m) {
        std::lock_guard<std::mutex> lk(m_);
        schedule_ = m;
    }

    PhaseGate() {
        ApiGuard api;
        for (auto& sl : ring_)
            for (auto& e : sl.ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    }
    ~PhaseGate() {
        ApiGuard api;
        for (auto& sl : ring_)
            for (auto& e : sl.ev) (void)hipEventDestroy(e);
    }
    PhaseGate(const PhaseGate&) = delete;
    PhaseGate& operator=(const PhaseGate&) = delete;

    // Announce a batch and make `s` wait for what its pyramid must not overlap.  Returns the ticket.
    long long begin_batch(hipStream_t s) {
        std::unique_lock<std::mutex> lk(m_);
        const long long g = next_++;
        Slot& me = slot(g);
        // the slot's previous user (g - kRing) finished long ago: every batch is waited for by its owner before the
        // owner submits again, and at most kRing / 2 contexts share a gate
        me.g = g;
        for (bool& r : me.rec) r = false;
        me.entered_c = me.entered_d = false;
        cv_.notify_all();
        if (g >= 1 && slot(g - 1).g == g - 1) {
            Slot& pred = slot(g - 1);
            if (schedule_ == 1 ? pred.entered_d : pred.entered_c) {
                cv_.wait(lk, [&] { return pred.rec[kD]; });
                (void)hipStreamWaitEvent(s, pred.ev[kD], 0);
            } else {
                cv_.wait(lk, [&] { return pred.rec[kE]; });
                (void)hipStreamWaitEvent(s, pred.ev[kE], 0);
            }
        }
        if (g >= 2 && slot(g - 2).g == g - 2) {
            Slot& pp = slot(g - 2);
            cv_.wait(lk, [&] { return pp.rec[kD]; });
            (void)hipStreamWaitEvent(s, pp.ev[kD], 0);
        }
        return g;
    }

    // Phase `ph` of batch g is complete once everything queued on `s` so far has run.
    void mark(long long g, Phase ph, hipStream_t s) {
        std::lock_guard<std::mutex> lk(m_);
        Slot& me = slot(g);
        if (me.g != g || me.rec[ph]) return;
        (void)hipEventRecord(me.ev[ph], s);
        me.rec[ph] = true;
        cv_.notify_all();
    }

    // Batch g is about to queue its cleanup chain on `s` (and on streams forked from `s` afterwards).
    void before_cleanup(long long g, hipStream_t s) {
        std::unique_lock<std::mutex> lk(m_);
        Slot& me = slot(g);
        if (me.g != g) return;
        if (schedule_ == 1) {   // the chain starts at once, beside the successor's pyramid
            me.entered_c = true;
            cv_.notify_all();
            return;
        }
        // The host runs far ahead of the device, so "is there a successor" is asked at DEVICE time: the question stays
        // open until the device has finished E(g).  A successor announced by then gets its pyramid in first; after
        // that point waiting could only leave the chip idle.
        while (next_ <= g + 1 && me.rec[kE] && hipEventQuery(me.ev[kE]) == hipErrorNotReady)
            cv_.wait_for(lk, std::chrono::microseconds(20));
        if (next_ > g + 1) {   // the successor exists: let its pyramid finish first
            Slot& succ = slot(g + 1);
            cv_.wait(lk, [&] { return succ.g == g + 1 && succ.rec[kP]; });
            (void)hipStreamWaitEvent(s, succ.ev[kP], 0);
        }
        me.entered_c = true;
        cv_.notify_all();
    }

    // Schedule 1: batch g is about to queue its descriptor stage on `s`, after its cleanup chain.  A successor announced by
    // the time the device finishes the chain gets its pyramid in first (the descriptors then share the chip with its
    // extrema / gradient pass: issue-bound beside bandwidth-bound); otherwise the descriptors start at once and the next
    // pyramid waits for them.
    void before_descriptors(long long g, hipStream_t s) {
        std::unique_lock<std::mutex> lk(m_);
        Slot& me = slot(g);
        if (me.g != g || schedule_ != 1) return;
        if (!me.rec[kC]) {
            (void)hipEventRecord(me.ev[kC], s);
            me.rec[kC] = true;
        }
        while (next_ <= g + 1 && hipEventQuery(me.ev[kC]) == hipErrorNotReady) cv_.wait_for(lk, std::chrono::microseconds(20));
        if (next_ > g + 1) {
            Slot& succ = slot(g + 1);
            cv_.wait(lk, [&] { return succ.g == g + 1 && succ.rec[kP]; });
            (void)hipStreamWaitEvent(s, succ.ev[kP], 0);
        }
        me.entered_d = true;
        cv_.notify_all();
    }

    // Whatever batch g has not marked yet is marked now (normal end, early return or exception).
    void finish(long long g, hipStream_t s) {
        std::lock_guard<std::mutex> lk(m_);
        Slot& me = slot(g);
        if (me.g != g) return;
        for (int ph = 0; ph < 4; ++ph)
            if (!me.rec[ph]) {
                (void)hipEventRecord(me.ev[ph], s);
                me.rec[ph] = true;
            }
        me.entered_c = me.entered_d = true;
        cv_.notify_all();
    }

    static constexpr int kRing = 8;
    // every batch is waited for by its owner before the owner submits again, so with at most kRing / 2 contexts a slot's
    // previous user (ticket g - kRing) is long done when ticket g takes it; sift_hip_set_gate enforces the limit
    static constexpr int kMaxContexts = kRing / 2;

private:
    struct Slot {
        long long g = -1;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        bool rec[4] = {false, false, false, false};
        bool entered_c = false, entered_d = false;
    };
    Slot& slot(long long g) { return ring_[g % kRing]; }
    std::mutex m_;
    std::condition_variable cv_;
    long long next_ = 0;
    int schedule_ = 1;
    Slot ring_[kRing];
};

}  // namespace sift_hip
