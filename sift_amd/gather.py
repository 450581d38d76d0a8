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
