"""Gather of per-rank keypoint lists on one rank: images are sharded across
ranks with no data-path collective; only the keypoint records and descriptors travel — an
all-gather of counts followed by direct point-to-point sends to the destination rank (RCCL over
xGMI with backend "nccl"; "gloo" on CPU for tests)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def gather_keypoints(kp_bytes: torch.Tensor, desc: torch.Tensor, counts: torch.Tensor, dst: int = 0):
    """kp_bytes: uint8 [total*20], desc: float32 [total*128], counts: int32 [n_local_images], all on
    this rank's device.  Returns on `dst` (kp_bytes_all, desc_all, counts_all) concatenated in rank
    order (= global image order for block sharding); None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = kp_bytes.device
    total = int(counts.sum().item())
    n_img = torch.tensor([counts.numel(), total], dtype=torch.int64, device=dev)
    sizes = [torch.empty_like(n_img) for _ in range(world)]
    dist.all_gather(sizes, n_img)
    sizes = [tuple(int(v) for v in s.tolist()) for s in sizes]
    # per-image counts: fixed-size all_gather needs equal shapes; pad to the largest image count
    m = max(s[0] for s in sizes)
    padded = torch.zeros(m, dtype=torch.int32, device=dev)
    padded[:counts.numel()] = counts.to(torch.int32)
    allc = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(allc, padded)
    if rank == dst:
        kps, descs, ops = [None] * world, [None] * world, []
        for r in range(world):
            t = sizes[r][1]
            if r == rank:
                kps[r], descs[r] = kp_bytes[:t * 20], desc[:t * 128]
            else:
                kps[r] = torch.empty(t * 20, dtype=torch.uint8, device=dev)
                descs[r] = torch.empty(t * 128, dtype=torch.float32, device=dev)
                if t:
                    ops += [dist.P2POp(dist.irecv, kps[r], r), dist.P2POp(dist.irecv, descs[r], r)]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        counts_all = torch.cat([allc[r][:sizes[r][0]] for r in range(world)])
        return torch.cat(kps), torch.cat(descs), counts_all
    if total:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, kp_bytes[:total * 20].contiguous(), dst),
                                         dist.P2POp(dist.isend, desc[:total * 128].contiguous(), dst)]):
            w.wait()
    return None


def pack_descriptors(desc: torch.Tensor) -> torch.Tensor:
    """128 -> 112 floats per keypoint for the wire: bin 7 of each of the 16 cells is never written by the
    reference (`% 7`, algorithms.cpp:135-150) and normalises to +0.0f, so it carries no information.
    Lossless: `unpack_descriptors(pack_descriptors(d))` is bit-identical to `d`."""
    return desc.view(-1, 16, 8)[:, :, :7].contiguous().view(-1)


def unpack_descriptors(packed: torch.Tensor) -> torch.Tensor:
    out = torch.zeros(packed.numel() // 112, 16, 8, dtype=packed.dtype, device=packed.device)
    out[:, :, :7] = packed.view(-1, 16, 7)
    return out.view(-1)


SPARSE_MASK_BYTES = 14   # 112 bits: which of the 112 informative floats of a descriptor are not +0.0f


def pack_sparse(desc: torch.Tensor):
    """128 floats per keypoint -> (mask bytes uint8 [K*14], values float32 [nnz]).  A descriptor cell only has mass in
    the bins its 16 samples fall into, so on real frames about a third of the 112 informative floats are set (37 % on
    the bench frames): the wire carries one presence bit per float and the set floats only — 14 + 4 * ~41 bytes instead
    of 448.  Presence is decided on the bit pattern (anything but +0.0f is sent, -0.0f and NaNs included), so
    `unpack_sparse(*pack_sparse(d))` is bit-identical to `d` whenever bin 7 of every cell is +0.0f (see pack_descriptors)."""
    d = desc.view(-1, 16, 8)[:, :, :7].reshape(-1, 112)
    nz = d.view(torch.int32) != 0
    values = d[nz]
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=desc.device)
    masks = (nz.view(-1, SPARSE_MASK_BYTES, 8).to(torch.int32) * weights).sum(-1).to(torch.uint8)
    return masks.reshape(-1), values


def unpack_sparse(masks: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
    m = masks.view(-1, SPARSE_MASK_BYTES).to(torch.int32)
    shifts = torch.arange(8, dtype=torch.int32, device=masks.device)
    nz = ((m.unsqueeze(-1) >> shifts) & 1).bool().view(-1, 112)
    d = torch.zeros(nz.shape, dtype=values.dtype, device=values.device)
    d[nz] = values
    return unpack_descriptors(d.view(-1))


def join_records(kp_bytes: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
    """20-byte keypoint records + 14-byte presence masks -> one 34-byte record per keypoint (one transfer instead of two)."""
    return torch.cat([kp_bytes.view(-1, 20), masks.view(-1, SPARSE_MASK_BYTES)], dim=1).reshape(-1)


def split_records(rec: torch.Tensor):
    r = rec.view(-1, 20 + SPARSE_MASK_BYTES)
    return r[:, :20].reshape(-1), r[:, 20:].reshape(-1)


class _DevArray:
    """Zero-copy view of device memory owned by libsift_hip (through `__cuda_array_interface__`)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_results(ctx, total: int, device, packed: bool = True, wire: str | None = None, rec_out: torch.Tensor | None = None,
                   defer_pack: bool = False):
    """The context's current results as torch tensors on `device`, in a wire format:
        "full"    (records uint8 [total*20], descriptors float32 [total*128])
        "packed"  (records uint8 [total*20], descriptors float32 [total*112])            see pack_descriptors
        "sparse"  (records uint8 [total*34] = record + presence mask, set floats [nnz])   see pack_sparse
    (`packed=True/False` selects "packed"/"full" when `wire` is not given.)  The library's arrays are read in place
    (no staging copy) by copies on torch's current stream, which this function waits for: nothing orders torch's stream
    against the library's own (non-blocking) streams, so the context must not start its next batch before the copies are
    done.  On return the tensors own their data and the context's slot may be released.
    defer_pack (wire "sparse"): the pack kernel is only queued (Context.sparse_pack(wait=False)); the context may start its next
    batch at once and the tensors are complete after `ctx.pack_wait()`, which any thread may call."""
    wire = wire or ("packed" if packed else "full")
    if total == 0:
        return (rec_out if rec_out is not None and rec_out.numel() == 0 else torch.empty(0, dtype=torch.uint8, device=device),
                torch.empty(0, dtype=torch.float32, device=device))
    kp_ptr, desc_ptr = ctx.result_device_ptrs()
    kp = torch.as_tensor(_DevArray(kp_ptr, total * 20), device=device)
    d = torch.as_tensor(_DevArray(desc_ptr, total * 512), device=device).view(torch.float32)
    if wire == "sparse":   # packed by the library's own kernels (kernels_wire.hip), straight into torch's memory
        nnz = ctx.sparse_size()
        # rec_out: where the records go (KeypointGather.records_buffer: the message buffer itself, no copy later)
        rec = rec_out if rec_out is not None else torch.empty(total * (20 + SPARSE_MASK_BYTES), dtype=torch.uint8, device=device)
        assert rec.numel() == total * (20 + SPARSE_MASK_BYTES) and rec.dtype == torch.uint8
        values = torch.empty(nnz, dtype=torch.float32, device=device)
        torch.cuda.current_stream(device).synchronize()   # the allocator may hand out memory still in use on torch's stream
        ctx.sparse_pack(rec.data_ptr(), values.data_ptr(), wait=not defer_pack)
        return rec, values
    out = kp.clone(), (pack_descriptors(d) if wire == "packed" else d.clone())
    torch.cuda.current_stream(device).synchronize()   # the library may rewrite its arrays as soon as the slot is released
    return out


class KeypointGather:
    """Gather of every step's keypoint lists on rank `dst` with NO per-step collective and no size exchange of its own.

    What rank `dst` must know before it can post a receive is how many bytes are coming.  Instead of exchanging sizes first
    (a blocking all_gather and a device-to-host read per step), a rank's message of step k carries, in one buffer,
        [ header(k): keypoints, record bytes, descriptor floats and per-image counts of ITS step k | records of step k-1 ]
    followed by a second message with the descriptor floats of step k-1 — so the sizes of a payload always arrive one step
    before the payload, and every receive is posted with its exact size.  `dst` reads a header (a few hundred bytes, long
    arrived) at the NEXT push; results therefore come out two pushes after they went in, and `flush` drains the rest.
    Point-to-point only (RCCL over xGMI: every rank has its own link to `dst`); the one all_gather is at
    construction (images per rank).  All ranks must call `push` the same number of times, then `flush` once.
    """

    def __init__(self, n_images_local: int, device, dst: int = 0, loopback: bool = False, concat: bool = True):
        """loopback (a world of ONE process, testing the transport itself on a one-GPU box): the process plays two ranks of the
        protocol — rank 0, the receiver, with no images of its own, and rank 1, the sender of its lists — and every message
        really goes through the backend's point-to-point path, to itself (sends and receives of a round in one group, as RCCL
        requires of a send to oneself)."""
        self.world, self.rank, self.dst, self.dev = dist.get_world_size(), dist.get_rank(), dst, device
        self.loopback = bool(loopback)
        # concat=False: a completed step is handed out as (list of per-rank records tensors, list of per-rank values tensors,
        # counts) in rank order - no copy on `dst`, whose own lists and every arrival stay where they are (with eight ranks the
        # concatenation is 0.17 GB of records and the own-values copy 0.1 GB per step on the one rank that also takes seven receives)
        self.concat = bool(concat)
        t = torch.tensor([n_images_local], dtype=torch.int64, device=device)
        allt = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(allt, t)                       # once, at construction
        self.n_img = [int(v.item()) for v in allt]
        if self.loopback:
            if self.world != 1:
                raise ValueError("loopback is a single-process mode")
            self.world, self.rank, self.dst, self.n_img = 2, 0, 0, [0, self.n_img[0]]
            self._lb_keep = []
        self.hdr_words = 3 + max(self.n_img)
        self.hdr_bytes = 8 * self.hdr_words
        self.step = 0
        self._rec_bufs = {}            # records_buffer: data_ptr of a records view -> the allocation with header room in front
        import threading
        self._rec_lock = threading.Lock()   # records_buffer may be called by the threads that pack, push by the one that gathers
        self.wire_bytes = 0            # received (dst) or sent (others), payload and headers
        self.wait_s = 0.0              # host time spent waiting for transfers / reading headers
        # sender side
        self._prev = None              # (records, values) of the previous push: they ride with the next header
        self._sends = []               # works + the tensors they read
        # receiver side
        self._own = {}                 # step -> (records, values, counts) of dst itself
        self._sizes = {}               # step -> {src: (n_rec_bytes, n_val_floats, counts)}   (from headers)
        self._msgs = None              # in flight: (step k of the headers, works, {src: A buffer}, values_all, rec sizes of step k-1)
        self._assembled = {}           # step -> values_all whose remote slices are complete once the message carrying them is

    # ---- helpers -------------------------------------------------------------------------------------------------
    def _header(self, total, n_rec, n_val, counts):
        h = torch.zeros(self.hdr_words, dtype=torch.int64)
        h[0], h[1], h[2] = total, n_rec, n_val
        if counts is not None and len(counts):
            h[3:3 + len(counts)] = torch.as_tensor(counts, dtype=torch.int64)
        return h.view(torch.uint8).to(self.dev)

    def _timed_wait(self, works):
        import time
        t0 = time.perf_counter()
        for w in works:
            w.wait()
        self.wait_s += time.perf_counter() - t0

    def _peer(self, r):
        return 0 if self.loopback else r

    def records_buffer(self, nbytes: int) -> torch.Tensor:
        """A records tensor of `nbytes` bytes with room for a header in front of it in the same allocation: pushed later, it
        is sent as it stands (the next step's header is written into that room) instead of being copied behind a header."""
        if nbytes <= 0:
            return torch.empty(0, dtype=torch.uint8, device=self.dev)
        buf = torch.empty(self.hdr_bytes + nbytes, dtype=torch.uint8, device=self.dev)
        view = buf[self.hdr_bytes:]
        with self._rec_lock:
            self._rec_bufs[view.data_ptr()] = buf
            while len(self._rec_bufs) > 6:     # a view that was staged elsewhere instead of being pushed as it is must not pin its buffer for ever
                self._rec_bufs.pop(next(iter(self._rec_bufs)))
        return view

    def _send_ops(self, total, records, values, counts):
        """message = [header of this step | records of the previous step], then the previous step's values"""
        prev_rec, prev_val = self._prev if self._prev is not None else (torch.empty(0, dtype=torch.uint8, device=self.dev), torch.empty(0, dtype=torch.float32, device=self.dev))
        head = self._header(total, 0 if records is None else records.numel(), 0 if values is None else values.numel(), counts)
        with self._rec_lock:
            a = self._rec_bufs.pop(prev_rec.data_ptr(), None) if prev_rec.numel() else None
        if a is not None and a.numel() == self.hdr_bytes + prev_rec.numel():
            a[:self.hdr_bytes].copy_(head)           # the message buffer was laid out by records_buffer
        else:
            a = torch.cat([head, prev_rec])
        ops = [dist.P2POp(dist.isend, a, self._peer(self.dst))]
        keep = [a]
        if prev_val.numel():
            ops.append(dist.P2POp(dist.isend, prev_val, self._peer(self.dst)))
            keep.append(prev_val)
        if not self.loopback:              # (the receiving side counts the same bytes)
            self.wire_bytes += a.numel() + 4 * prev_val.numel()
        self._prev = (records, values) if records is not None else None
        return ops, keep

    def _send(self, total, records, values, counts):
        """non-dst ranks"""
        self._timed_wait([w for w, _ in self._sends])
        self._sends = []
        ops, keep = self._send_ops(total, records, values, counts)
        for w in dist.batch_isend_irecv(ops):
            self._sends.append((w, keep))

    def _receive_round(self, k, have_own, extra_ops=()):
        """dst, at push k (or the flush rounds): finish the messages of round k-1 (headers of step k-1, payload of step k-2),
        emit step k-2, post the receives of round k (headers of step k, payload of step k-1)."""
        import time
        out = []
        others = [r for r in range(self.world) if r != self.dst]
        if self._msgs is not None:
            hk, works, bufs, rec_sizes = self._msgs
            self._timed_wait(works)
            t0 = time.perf_counter()
            heads = torch.stack([bufs[r][:self.hdr_bytes] for r in others]).cpu().view(torch.int64).reshape(len(others), self.hdr_words)
            self.wait_s += time.perf_counter() - t0
            self._sizes[hk] = {r: (int(heads[i, 1]), int(heads[i, 2]), heads[i, 3:3 + self.n_img[r]].to(torch.int32), int(heads[i, 0])) for i, r in enumerate(others)}
            if hk - 1 in self._assembled:      # the payload of step hk-1 rode in these messages
                out.append(self._finish(hk - 1, bufs, rec_sizes))
            self._msgs = None
        # receives of round k: payload sizes are those of step k-1
        prev = self._sizes.get(k - 1)
        rec_sizes = {r: (prev[r][0] if prev else 0) for r in others}
        bufs = {r: torch.empty(self.hdr_bytes + rec_sizes[r], dtype=torch.uint8, device=self.dev) for r in others}
        ops = list(extra_ops) + [dist.P2POp(dist.irecv, bufs[r], self._peer(r)) for r in others]
        if prev:
            own_rec, own_val, own_cnt = self._own[k - 1]
            n_val = {r: prev[r][1] for r in others}
            n_val[self.dst] = own_val.numel()
            offs, run = {}, 0
            for r in range(self.world):
                offs[r] = run
                run += n_val[r]
            if self.concat:
                vals = torch.empty(run, dtype=torch.float32, device=self.dev)
                vals[offs[self.dst]:offs[self.dst] + n_val[self.dst]].copy_(own_val)
                for r in others:
                    if n_val[r]:
                        ops.append(dist.P2POp(dist.irecv, vals[offs[r]:offs[r] + n_val[r]], self._peer(r)))
                self._assembled[k - 1] = vals
            else:
                parts = [own_val if r == self.dst else torch.empty(n_val[r], dtype=torch.float32, device=self.dev) for r in range(self.world)]
                for r in others:
                    if n_val[r]:
                        ops.append(dist.P2POp(dist.irecv, parts[r], self._peer(r)))
                self._assembled[k - 1] = parts
            self.wire_bytes += sum(rec_sizes.values()) + 4 * sum(n_val[r] for r in others)
        self.wire_bytes += self.hdr_bytes * len(others)
        works = dist.batch_isend_irecv(ops) if ops else []
        self._msgs = (k, works, bufs, rec_sizes)
        return out

    def _finish(self, j, bufs, rec_sizes):
        """records of step j arrived behind the headers in `bufs`, its values straight into their slices"""
        own_rec, own_val, own_cnt = self._own.pop(j)
        sizes = self._sizes.pop(j)
        parts, counts = [], []
        for r in range(self.world):
            if r == self.dst:
                parts.append(own_rec)
                counts.append(torch.as_tensor(own_cnt, dtype=torch.int32))
            else:
                parts.append(bufs[r][self.hdr_bytes:self.hdr_bytes + rec_sizes[r]])
                counts.append(sizes[r][2])
        if not self.concat:
            return parts, self._assembled.pop(j), torch.cat(counts)
        return torch.cat(parts), self._assembled.pop(j), torch.cat(counts)

    # ---- API ---------------------------------------------------------------------------------------------------------
    def push(self, records: torch.Tensor, values: torch.Tensor, counts):
        """Hand over this rank's lists of one step (records uint8, descriptor values float32, per-image counts).  Returns the
        list of steps completed on `dst` by this call, each (records_all, values_all, counts_all) in rank order (= global
        image order for block sharding; with concat=False records_all / values_all are lists of one tensor per rank instead of
        one concatenated tensor); always [] on the other ranks.  The tensors must stay untouched until two pushes (or
        the flush) later."""
        k = self.step
        self.step += 1
        total = int(sum(int(c) for c in counts))
        if self.loopback:
            ops, keep = self._send_ops(total, records, values, counts)          # the sender's half of round k ...
            self._lb_keep = self._lb_keep[-1:] + [keep]                         # (alive until the round after is posted)
            self._own[k] = (torch.empty(0, dtype=torch.uint8, device=self.dev), torch.empty(0, dtype=torch.float32, device=self.dev), [])
            return self._receive_round(k, True, extra_ops=ops)                  # ... in one group with the receiver's
        if self.world == 1:
            c = torch.as_tensor(counts, dtype=torch.int32)
            return [(records, values, c)] if self.concat else [([records], [values], c)]
        if self.rank != self.dst:
            self._send(total, records, values, counts)
            return []
        self._own[k] = (records, values, counts)
        return self._receive_round(k, True)

    def flush(self):
        """After the last push: moves the payloads still on their way.  Returns the remaining completed steps on `dst`."""
        if self.world == 1:
            return []
        k = self.step
        extra = ()
        if self.loopback:
            extra, keep = self._send_ops(-1, None, None, None)
            self._lb_keep = self._lb_keep[-1:] + [keep]
        elif self.rank != self.dst:
            self._send(-1, None, None, None)      # an end header carrying the last step's payload
            self._timed_wait([w for w, _ in self._sends])
            self._sends = []
            return []
        out = self._receive_round(k, False, extra_ops=extra)        # finishes round k-1, posts round k (payload of step k-1)
        if self._msgs is not None:
            hk, works, bufs, rec_sizes = self._msgs
            self._timed_wait(works)
            if hk - 1 in self._assembled:
                self._sizes.setdefault(hk, None)
                out.append(self._finish(hk - 1, bufs, rec_sizes))
            self._msgs = None
        return out


class GatherThread:
    """The one thread of a rank that talks to the gather.  The host threads that run batches (BatchPipeline.run_stream: one
    per context, each taking its next step itself) hand their packed lists to `put(seq, records, values, counts)` and go on;
    this thread pushes them to the KeypointGather in step order `seq` = 0, 1, 2, ... (the batches of two contexts may finish
    out of order), so that a rank's point-to-point messages are issued by one thread in the same order on every rank, and
    no batch waits for a transfer.  `close()` flushes the gather, joins the thread and re-raises what it raised.
    `on_done(list of (records_all, values_all, counts_all))` is called on this thread with the steps completed on `dst`."""

    def __init__(self, gatherer: "KeypointGather", on_done=None, cuda_device: int | None = None, keep: int = 3):
        import queue
        import threading
        self._g, self._on_done, self._dev, self._keep_n = gatherer, on_done, cuda_device, keep
        self._q = queue.Queue()
        self._error = None
        self._t = threading.Thread(target=self._run, name="sift-gather")
        self._t.start()

    def _run(self):
        closed = False
        try:
            if self._dev is not None:
                torch.cuda.set_device(self._dev)
            pending, want, keep = {}, 0, []
            while True:
                got = self._q.get()
                if got is None:
                    closed = True
                    break
                pending[got[0]] = got[1:]
                while want in pending:
                    rec, val, counts, ready = pending.pop(want)
                    if ready is not None:
                        ready()
                    keep.append((rec, val))        # the gather reads them until two pushes later
                    del keep[:-self._keep_n]
                    done = self._g.push(rec, val, counts)
                    if self._on_done is not None and done:
                        self._on_done(done)
                    want += 1
            if pending:
                raise RuntimeError(f"gather thread closed with steps {sorted(pending)} waiting for step {want}")
            done = self._g.flush()
            if self._on_done is not None and done:
                self._on_done(done)
        except BaseException as e:   # noqa: BLE001
            self._error = e
            while not closed:           # keep draining so that no producer blocks; close() reports the error
                closed = self._q.get() is None

    def put(self, seq: int, records, values, counts, ready=None) -> None:
        """`ready`, if given, is called on the gather thread right before the push (e.g. Context.pack_wait of a deferred pack)."""
        self._q.put((seq, records, values, counts, ready))

    def close(self) -> None:
        self._q.put(None)
        self._t.join()
        if self._error is not None:
            raise self._error


class GatherHandle:
    """One gather in flight: the point-to-point works plus what `finish` needs to assemble the result."""

    def __init__(self, works, parts, sizes, allc, keep):
        self.works, self.parts, self.sizes, self.allc, self.keep = works, parts, sizes, allc, keep


def gather_start(kp_bytes: torch.Tensor, desc: torch.Tensor, counts: torch.Tensor, dst: int = 0, floats_per_kp: int | None = 128,
                 bytes_per_kp: int = 20) -> GatherHandle:
    """Non-blocking form of `gather_keypoints`: the (tiny) count exchange is done here, the record and
    descriptor transfers are only STARTED, so they overlap whatever the caller does next (the next
    batch's kernels run on the library's own streams).  The caller must not touch `kp_bytes` / `desc`
    until `gather_finish` returned.  `bytes_per_kp` / `floats_per_kp` describe the wire format (20 / 128 full,
    20 / 112 packed, 34 / None sparse: a variable number of floats, `desc.numel()` of them)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = kp_bytes.device
    total = int(counts.sum().item())
    n_rec = total * bytes_per_kp
    n_flt = int(desc.numel()) if floats_per_kp is None else total * floats_per_kp
    n_img = torch.tensor([counts.numel(), total, n_rec, n_flt], dtype=torch.int64, device=dev)
    sizes = [torch.empty_like(n_img) for _ in range(world)]
    dist.all_gather(sizes, n_img)
    sizes = [tuple(int(v) for v in s.tolist()) for s in sizes]
    m = max(s[0] for s in sizes)
    padded = torch.zeros(m, dtype=torch.int32, device=dev)
    padded[:counts.numel()] = counts.to(torch.int32)
    allc = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(allc, padded)
    works, parts = [], None
    if rank == dst:
        # one contiguous destination in rank order (= global image order): every rank's part is received
        # straight into its slice, the local part is copied there
        kp_all = torch.empty(sum(sz[2] for sz in sizes), dtype=torch.uint8, device=dev)
        desc_all = torch.empty(sum(sz[3] for sz in sizes), dtype=torch.float32, device=dev)
        ops, ro, fo = [], 0, 0
        for r in range(world):
            nr, nf = sizes[r][2], sizes[r][3]
            if r == rank:
                kp_all[ro:ro + nr].copy_(kp_bytes[:nr])
                desc_all[fo:fo + nf].copy_(desc[:nf])
            else:
                if nr:
                    ops.append(dist.P2POp(dist.irecv, kp_all[ro:ro + nr], r))
                if nf:
                    ops.append(dist.P2POp(dist.irecv, desc_all[fo:fo + nf], r))
            ro += nr
            fo += nf
        if ops:
            works = dist.batch_isend_irecv(ops)
        parts = (kp_all, desc_all)
        keep = (kp_bytes, desc)
    else:
        keep = (kp_bytes[:n_rec].contiguous(), desc[:n_flt].contiguous())
        ops = []
        if n_rec:
            ops.append(dist.P2POp(dist.isend, keep[0], dst))
        if n_flt:
            ops.append(dist.P2POp(dist.isend, keep[1], dst))
        if ops:
            works = dist.batch_isend_irecv(ops)
    return GatherHandle(works, parts, sizes, allc, keep)


def gather_finish(h: GatherHandle):
    """Wait for the transfers of `h`.  On the destination rank returns (kp_bytes_all, desc_all, counts_all)
    in rank order; None on the other ranks."""
    for w in h.works:
        w.wait()
    if h.parts is None:
        return None
    world = len(h.sizes)
    counts_all = torch.cat([h.allc[r][:h.sizes[r][0]] for r in range(world)])
    return h.parts[0], h.parts[1], counts_all
