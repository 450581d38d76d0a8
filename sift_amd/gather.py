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


def device_results(ctx, total: int, device, packed: bool = True, wire: str | None = None):
    """The context's current results as torch tensors on `device`, in a wire format:
        "full"    (records uint8 [total*20], descriptors float32 [total*128])
        "packed"  (records uint8 [total*20], descriptors float32 [total*112])            see pack_descriptors
        "sparse"  (records uint8 [total*34] = record + presence mask, set floats [nnz])   see pack_sparse
    (`packed=True/False` selects "packed"/"full" when `wire` is not given.)  The library's arrays are read in place
    (no staging copy); the copies are queued on torch's current stream, so the context may start its next batch right
    away (its result arrays are only rewritten by the descriptor kernel at the end of that batch)."""
    wire = wire or ("packed" if packed else "full")
    if total == 0:
        return (torch.empty(0, dtype=torch.uint8, device=device), torch.empty(0, dtype=torch.float32, device=device))
    kp_ptr, desc_ptr = ctx.result_device_ptrs()
    kp = torch.as_tensor(_DevArray(kp_ptr, total * 20), device=device)
    d = torch.as_tensor(_DevArray(desc_ptr, total * 512), device=device).view(torch.float32)
    if wire == "sparse":   # packed by the library's own kernels (kernels_wire.hip), straight into torch's memory
        nnz = ctx.sparse_size()
        rec = torch.empty(total * (20 + SPARSE_MASK_BYTES), dtype=torch.uint8, device=device)
        values = torch.empty(nnz, dtype=torch.float32, device=device)
        torch.cuda.current_stream(device).synchronize()   # the allocator may hand out memory still in use on torch's stream
        ctx.sparse_pack(rec.data_ptr(), values.data_ptr())
        return rec, values
    return kp.clone(), (pack_descriptors(d) if wire == "packed" else d.clone())


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
