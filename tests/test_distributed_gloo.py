"""World-size-2 gloo test of the keypoint gather used for N > 1 (no GPU needed)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sift_amd.gather import gather_keypoints


def _fake_rank_data(rank):
    rng = np.random.default_rng(100 + rank)
    counts = np.array([3, 0, 5] if rank == 0 else [0, 4], np.int32)
    total = int(counts.sum())
    kp = rng.integers(0, 256, total * 20, dtype=np.uint8)
    desc = rng.random(total * 128).astype(np.float32)
    return counts, kp, desc


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    counts, kp, desc = _fake_rank_data(rank)
    out = gather_keypoints(torch.from_numpy(kp), torch.from_numpy(desc), torch.from_numpy(counts), dst=0)
    if rank == 0:
        q.put((out[0].numpy().copy(), out[1].numpy().copy(), out[2].numpy().copy()))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_two_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    kp, desc, counts = q.get(timeout=120)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    c0, k0, d0 = _fake_rank_data(0)
    c1, k1, d1 = _fake_rank_data(1)
    assert (counts == np.concatenate([c0, c1])).all()
    assert (kp == np.concatenate([k0, k1])).all()
    assert (desc == np.concatenate([d0, d1])).all()
