"""World-size-2 gloo test of the keypoint gather used for N > 1 (no GPU needed)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sift_amd.gather import (gather_finish, gather_keypoints, gather_start, join_records, pack_descriptors, pack_sparse,
                             split_records, unpack_descriptors, unpack_sparse)


def _fake_rank_data(rank):
    rng = np.random.default_rng(100 + rank)
    counts = np.array([3, 0, 5] if rank == 0 else [0, 4], np.int32)
    total = int(counts.sum())
    kp = rng.integers(0, 256, total * 20, dtype=np.uint8)
    desc = rng.random(total * 128).astype(np.float32)
    return counts, kp, desc


def _sparsify(desc):
    """Descriptor-like data: bin 7 of every cell +0.0f, most other bins too, a -0.0f and a NaN among the set ones."""
    d = desc.reshape(-1, 16, 8).copy()
    rng = np.random.default_rng(d.size)
    d[rng.random(d.shape) < 0.6] = 0.0
    d[:, :, 7] = 0.0
    if d.shape[0]:
        d[0, 0, 0] = -0.0
        d[0, 1, 2] = np.nan
    return d.reshape(-1)


def test_sparse_wire_format_round_trip():
    _, kp, desc = _fake_rank_data(0)
    d = _sparsify(desc)
    masks, values = pack_sparse(torch.from_numpy(d))
    assert masks.numel() == (d.size // 128) * 14
    assert values.numel() == int((d.view(np.int32) != 0).sum())          # -0.0f and NaN are sent, +0.0f is not
    assert unpack_sparse(masks, values).numpy().tobytes() == d.tobytes()
    rec = join_records(torch.from_numpy(kp), masks)
    k2, m2 = split_records(rec)
    assert k2.numpy().tobytes() == kp.tobytes() and bool((m2 == masks).all())
    empty = pack_sparse(torch.zeros(0))
    assert empty[0].numel() == 0 and empty[1].numel() == 0 and unpack_sparse(*empty).numel() == 0


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
    # pipelined form (bench.py, N > 1): two gathers in flight, finished in order
    hs = []
    for step in range(3):
        c2, k2, d2 = _fake_rank_data(rank + 10 * (step + 1))
        if step == 1:   # packed wire format: 112 of 128 floats per keypoint
            d2 = d2.reshape(-1, 16, 8).copy()
            d2[:, :, 7] = 0.0
            d2 = d2.reshape(-1)
            packed = pack_descriptors(torch.from_numpy(d2))
            assert unpack_descriptors(packed).numpy().tobytes() == d2.tobytes()
            hs.append((step, gather_start(torch.from_numpy(k2), packed, torch.from_numpy(c2), dst=0, floats_per_kp=112)))
            continue
        if step == 2:   # sparse wire format: presence bits joined to the records, set floats only (ragged per rank)
            d2 = _sparsify(d2)
            masks, values = pack_sparse(torch.from_numpy(d2))
            hs.append((step, gather_start(join_records(torch.from_numpy(k2), masks), values, torch.from_numpy(c2), dst=0,
                                          floats_per_kp=None, bytes_per_kp=34)))
        else:
            hs.append((step, gather_start(torch.from_numpy(k2), torch.from_numpy(d2), torch.from_numpy(c2), dst=0)))
        if len(hs) > 2:
            st, h = hs.pop(0)
            res = gather_finish(h)
            if rank == 0:
                q.put((st, res[0].numpy().copy(), res[1].numpy().copy(), res[2].numpy().copy()))
    for st, h in hs:
        res = gather_finish(h)
        if rank == 0:
            q.put((st, res[0].numpy().copy(), res[1].numpy().copy(), res[2].numpy().copy()))
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
    piped = [q.get(timeout=120) for _ in range(3)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    c0, k0, d0 = _fake_rank_data(0)
    c1, k1, d1 = _fake_rank_data(1)
    assert (counts == np.concatenate([c0, c1])).all()
    assert (kp == np.concatenate([k0, k1])).all()
    assert (desc == np.concatenate([d0, d1])).all()
    for st, kp2, desc2, counts2 in piped:
        a0, b0, e0 = _fake_rank_data(0 + 10 * (st + 1))
        a1, b1, e1 = _fake_rank_data(1 + 10 * (st + 1))
        assert (counts2 == np.concatenate([a0, a1])).all()
        if st != 2:
            assert (kp2 == np.concatenate([b0, b1])).all()
        if st == 1:
            full = np.concatenate([e0, e1]).reshape(-1, 16, 8).copy()
            full[:, :, 7] = 0.0
            desc2 = unpack_descriptors(torch.from_numpy(desc2)).numpy()
            assert desc2.tobytes() == full.reshape(-1).tobytes()
            continue
        if st == 2:
            full = np.concatenate([_sparsify(e0), _sparsify(e1)])
            recs, masks = split_records(torch.from_numpy(kp2))
            assert recs.numpy().tobytes() == np.concatenate([b0, b1]).tobytes()
            assert unpack_sparse(masks, torch.from_numpy(desc2)).numpy().tobytes() == full.tobytes()
            continue
        assert (desc2 == np.concatenate([e0, e1])).all()


def _step_data(rank, step):
    """ragged per rank and step; rank 1 has nothing at step 2, everybody nothing at step 3"""
    rng = np.random.default_rng(1000 * step + rank)
    n_img = 2 + rank
    counts = rng.integers(0, 6, n_img).astype(np.int32)
    if (rank == 1 and step == 2) or step == 3:
        counts[:] = 0
    total = int(counts.sum())
    rec = rng.integers(0, 256, total * 34, dtype=np.uint8)
    val = rng.random(int(rng.integers(0, 40 * total + 1))).astype(np.float32)
    return counts, rec, val


def _kg_worker(rank, world, port, q, steps, concat=True):
    from sift_amd.gather import KeypointGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = KeypointGather(2 + rank, torch.device("cpu"), dst=0, concat=concat)
    done = []
    for step in range(steps):
        c, r, v = _step_data(rank, step)
        rt = torch.from_numpy(r)
        if step % 2 == 1:        # every other step: the records sit in a message buffer with header room (no copy at send time)
            rb = g.records_buffer(rt.numel())
            rb.copy_(rt)
            rt = rb
        done += g.push(rt, torch.from_numpy(v), c)
    done += g.flush()
    if rank == 0:
        if not concat:     # one tensor per rank and step, nothing copied on rank 0: its own lists are the tensors it pushed
            assert all(len(a) == world and len(b) == world for a, b, _ in done)
            done = [(torch.cat(list(a)), torch.cat(list(b)), cc) for a, b, cc in done]
        q.put([(a.numpy().copy(), b.numpy().copy(), cc.numpy().copy()) for a, b, cc in done] + [g.wire_bytes])
    else:
        assert done == []
    dist.barrier()
    dist.destroy_process_group()


def _run_kg(world, steps, concat=True):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_kg_worker, args=(r, world, port, q, steps, concat)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    wire = got.pop()
    assert len(got) == steps                    # every step comes out exactly once, in order
    want_bytes = 0
    for step, (rec, val, counts) in enumerate(got):
        parts = [_step_data(r, step) for r in range(world)]
        assert counts.tolist() == np.concatenate([p[0] for p in parts]).tolist()
        assert rec.tobytes() == np.concatenate([p[1] for p in parts]).tobytes()
        assert val.tobytes() == np.concatenate([p[2] for p in parts]).tobytes()
        want_bytes += sum(p[1].size + 4 * p[2].size for p in parts[1:])
    hdr = 8 * (3 + (2 + world - 1))
    assert wire == want_bytes + hdr * (world - 1) * (steps + 1)   # payload once, one header per message, nothing else


def test_keypoint_gather_without_per_step_collectives_two_ranks():
    """sift_amd.gather.KeypointGather: sizes ride one step ahead of the payload, so no all_gather / size read per step."""
    _run_kg(2, 5)


def test_keypoint_gather_three_ranks_single_step():
    _run_kg(3, 1)


def test_keypoint_gather_per_rank_lists_three_ranks():
    """concat=False (what bench.py uses): completed steps come out as one tensor per rank, in rank order."""
    _run_kg(3, 4, concat=False)


def test_keypoint_gather_eight_ranks():
    """The world size of the target node (8 GPUs, BASELINE config 4): eight processes, ragged lists, four steps; rank 0 ends up
    with every step's lists in rank order = global image order."""
    _run_kg(8, 4)


def _gt_worker(rank, world, port, q, steps):
    """bench.py's N > 1 host loop without a GPU: two threads (the contexts' host threads of BatchPipeline.run_stream) hand their
    steps' lists to the rank's GatherThread, deliberately out of order; the thread pushes them in step order."""
    import threading
    import time
    from sift_amd.gather import GatherThread, KeypointGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = KeypointGather(2 + rank, torch.device("cpu"), dst=0)
    done = []
    gt = GatherThread(g, done.extend)

    def producer(slot):
        for step in range(slot, steps, 2):
            c, r, v = _step_data(rank, step)
            if slot == 0:
                time.sleep(0.02 * ((rank + step) % 3))      # slot 1's steps often overtake slot 0's
            gt.put(step, torch.from_numpy(r), torch.from_numpy(v), c)

    ts = [threading.Thread(target=producer, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    gt.close()
    if rank == 0:
        q.put([(a.numpy().copy(), b.numpy().copy(), cc.numpy().copy()) for a, b, cc in done] + [g.wire_bytes])
    else:
        assert done == []
    dist.barrier()
    dist.destroy_process_group()


def test_gather_thread_pushes_in_step_order_eight_ranks():
    world, steps = 8, 5
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gt_worker, args=(r, world, port, q, steps)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got.pop()
    assert len(got) == steps
    for step, (rec, val, counts) in enumerate(got):
        parts = [_step_data(r, step) for r in range(world)]
        assert counts.tolist() == np.concatenate([p[0] for p in parts]).tolist()
        assert rec.tobytes() == np.concatenate([p[1] for p in parts]).tobytes()
        assert val.tobytes() == np.concatenate([p[2] for p in parts]).tobytes()


def test_gather_thread_reports_a_gap():
    """A step that never arrives is an error at close(), not a hang."""
    from sift_amd.gather import GatherThread

    class _G:
        def push(self, *a):
            return []

        def flush(self):
            return []

    gt = GatherThread(_G())
    gt.put(1, torch.zeros(0, dtype=torch.uint8), torch.zeros(0), [0])
    try:
        gt.close()
    except RuntimeError as e:
        assert "waiting for step 0" in str(e)
    else:
        raise AssertionError("no error for the missing step 0")


class _DoneWork:
    def wait(self):
        return True


def _fake_self_transport(ops):
    """what a backend does with one group of sends and receives to oneself: the i-th send lands in the i-th receive"""
    sends = [o.tensor for o in ops if o.op is dist.isend]
    recvs = [o.tensor for o in ops if o.op is dist.irecv]
    assert len(sends) == len(recvs), (len(sends), len(recvs))
    for a, b in zip(sends, recvs):
        assert a.numel() == b.numel() and a.dtype == b.dtype, (a.shape, b.shape)   # every receive was posted with its exact size
        b.copy_(a)
    return [_DoneWork()]


def test_keypoint_gather_loopback_protocol(monkeypatch):
    """KeypointGather(loopback=True): one process plays sender and receiver and every message goes through the backend's
    point-to-point path to itself (the form tests/test_gpu_parity.py runs over RCCL on the one-GPU box; gloo cannot send to
    itself, so the transport is a stand-in here and only the protocol — sizes one step ahead, one group per round — is checked)."""
    from sift_amd.gather import KeypointGather
    import sift_amd.gather as G
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(s.getsockname()[1])
    s.close()
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        monkeypatch.setattr(G.dist, "batch_isend_irecv", _fake_self_transport)
        g = KeypointGather(3, torch.device("cpu"), loopback=True)
        rng = np.random.default_rng(3)
        data, done = [], []
        for step in range(5):
            c = rng.integers(0, 6, 3) if step != 2 else np.zeros(3, np.int64)     # one step without a keypoint
            r = rng.integers(0, 256, int(c.sum()) * 34).astype(np.uint8)
            v = rng.random(int(c.sum()) * 9).astype(np.float32)
            data.append((c, r, v))
            done += g.push(torch.from_numpy(r), torch.from_numpy(v), c)
        done += g.flush()
        assert len(done) == 5
        for (c, r, v), (R, V, C) in zip(data, done):
            assert C.tolist() == c.tolist() and R.numpy().tobytes() == r.tobytes() and V.numpy().tobytes() == v.tobytes()
        assert g.wire_bytes == sum(r.size + 4 * v.size for _, r, v in data) + 8 * (3 + 3) * 6
    finally:
        dist.destroy_process_group()


def test_bench_launcher_reports_a_failing_rank():
    """`python bench.py --gpus 2` as a bare command starts its own rank processes (bench.py launch_ranks: RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set per child, before the parent has imported torch or touched a GPU) and exits non-zero when a rank
    fails - here every rank does, on a machine without a GPU, or because the library option does not exist - without printing a
    result line.  (The passing run is test_bench_starts_its_own_ranks, on the GPU.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-device", "--backend", "gloo", "--steps", "1",
                        "--warmup", "0", "--frames", "1", "--no-cpu-baseline", "--no-extras", "--set", "no_such_option=1"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "rank 0 exited with status" in r.stderr or "rank 1 exited with status" in r.stderr
    # under a launcher that has set WORLD_SIZE the script does not start ranks of its own: a mismatch is refused
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
