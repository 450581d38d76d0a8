"""Batches in flight on one GPU: a ring of contexts, each with its own streams and host thread.

One `Sift::calculate()` over a batch (sift.cpp:19-57) has two stretches that cannot fill the chip however the batch
is laid out: the cleanup steps (one workgroup per image, a chain of partition rounds) and the tails of the small
octaves.  They are filled by the NEXT batch's bandwidth-bound kernels when two batches are in flight, which a
single context cannot do (its result arrays and level buffers belong to one batch at a time).  This module is the
host-side scheduler for that: `depth` contexts, batch k goes to context k % depth, every context is driven by one
worker thread of its own (the C ABI blocks until a batch is done and releases the GIL meanwhile).

With `gated=True` (default) the contexts are joined by a phase gate (sift_amd/csrc/phase_gate.h): the device then
runs  ... | pyramid(k+1) || cleanup(k) | extrema(k+1) || descriptors(k) | pyramid(k+2) || cleanup(k+1) | ...  - the cleanup
chain, which cannot fill the chip, runs under the next batch's pyramid and the descriptors under its extrema / gradient
pass (library option `gate_schedule` = 0 brings back the order of rounds 1 - 2, in which a pyramid never shares the chip).
`gated=False` leaves the interleaving to the GPU's queues.

Results are those of the plain context, batch for batch — the contexts share nothing but the GPU.
"""
from __future__ import annotations

import threading
from concurrent.futures import Future, ThreadPoolExecutor

from .sift import Context, Gate


class Ticket:
    """One submitted batch.  `result()` waits for it and returns its context, which holds the results
    (`total()`, `counts()`, `results()`, `result_device_ptrs()`) until `release()` hands the slot back."""

    def __init__(self, pipe: "BatchPipeline", slot: int, future: Future):
        self._pipe, self.slot, self._future, self._released = pipe, slot, future, False

    def result(self) -> Context:
        self._future.result()          # re-raises what calculate raised (PreconditionViolation, ...)
        return self._pipe.contexts[self.slot]

    def release(self) -> None:
        if not self._released:
            self._released = True
            try:
                self._future.result()
            except Exception:
                pass                   # the caller has seen it through result(), or does not care
            self._pipe._busy[self.slot] = None


class BatchPipeline:
    def __init__(self, device: int = 0, depth: int = 2, options: dict | None = None, gated: bool = True):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.depth = depth
        self.contexts = [Context(device) for _ in range(depth)]
        self.gate = Gate(device) if (gated and depth > 1) else None
        for c in self.contexts:
            if self.gate is not None:
                c.set_gate(self.gate)
            for name, value in (options or {}).items():
                c.set_option(name, int(value))
        # one single-thread executor per context: batches of a context run in submission order
        self._workers = [ThreadPoolExecutor(1, thread_name_prefix=f"sift-pipe{i}") for i in range(depth)]
        self._busy: list[Ticket | None] = [None] * depth
        self._next = 0

    def _submit(self, method: str, *args, then=None) -> Ticket:
        slot = self._next
        if self._busy[slot] is not None:
            raise RuntimeError(f"pipeline slot {slot} still holds an unreleased batch: release() it before submitting batch number depth + 1")
        self._next = (slot + 1) % self.depth
        ctx = self.contexts[slot]

        def job():
            getattr(ctx, method)(*args)
            if then is not None:       # e.g. the download of the batch's lists: on the slot's own thread, beside the other slots' batches
                then(ctx, slot)

        t = Ticket(self, slot, self._workers[slot].submit(job))
        self._busy[slot] = t
        return t

    def submit_device(self, dev_ptr: int, n: int, w: int, h: int, params) -> Ticket:
        """Queue one batch of n device-resident w x h float frames.  Returns at once.  The slot's previous
        ticket must have been released (its results are overwritten by this batch)."""
        return self._submit("calculate_batch_device", dev_ptr, n, w, h, params)

    def submit(self, imgs, params, then=None) -> Ticket:
        """Same for a host array [n, h, w] float32 or uint8 (uploaded by the slot's worker thread).  `then(context, slot)`, if
        given, runs on that thread right after the batch (e.g. Context.results_sparse into the caller's buffers), so that a
        host that moves frames in and lists out keeps upload, kernels and download of different batches going side by side."""
        return self._submit("calculate_batch", imgs, params, then=then)

    def run_stream(self, source, sink=None) -> None:
        """Every slot's worker thread feeds itself: it takes the next batch from `source()` (called under a lock, in order; a
        tuple (dev_ptr, n, w, h, params) or (host_array, params); None ends the stream), runs it on its context and hands the
        context to `sink(context, slot, item)` before it takes the next one.  Returns when the stream has ended and every batch
        is done.  Compared with submit() + result() from one dispatching thread, no batch waits for that thread to wake up
        between the end of one batch of a context and the start of its next (two thread hand-overs, ~0.1 ms with the GIL: the
        device sat idle for them, profiles/r03_timeline_pipelined.txt)."""
        if any(t is not None for t in self._busy):
            raise RuntimeError("run_stream needs every slot free: release() the outstanding tickets first")
        lock = threading.Lock()
        ended = [False]

        def loop(slot):
            ctx = self.contexts[slot]
            try:
                while True:
                    with lock:
                        item = None if ended[0] else source()
                        if item is None:
                            ended[0] = True
                            return
                    if len(item) == 2:
                        ctx.calculate_batch(*item)
                    else:
                        ctx.calculate_batch_device(*item)
                    if sink is not None:
                        sink(ctx, slot, item)
            except BaseException:
                with lock:          # the other threads finish the batch they are on and take no further one
                    ended[0] = True
                raise

        futures = [w.submit(loop, i) for i, w in enumerate(self._workers)]
        first = None
        for f in futures:
            try:
                f.result()
            except Exception as e:
                first = first or e
        if first is not None:
            raise first

    def close(self) -> None:
        for t in self._busy:
            if t is not None:
                t.release()
        for w in self._workers:
            w.shutdown(wait=True)
        for c in self.contexts:
            c.close()
        self.contexts = []
        if self.gate is not None:
            self.gate.close()
            self.gate = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
