"""What RCCL does on a 1-GPU box: a world of ONE rank (init, all_gather, all_reduce, barrier) and a send to itself."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(8, device=dev, dtype=torch.int64)
out = [torch.zeros_like(x)]
dist.all_gather(out, x)
y = x.float()
dist.all_reduce(y)
dist.barrier()
torch.cuda.synchronize()
print("world-1 collectives ok", out[0].tolist(), y.tolist(), flush=True)
if "--self-send" in sys.argv:
    a = torch.ones(1024, device=dev)
    b = torch.zeros(1024, device=dev)
    ops = [dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    print("self send ok", float(b.sum()), flush=True)
dist.destroy_process_group()
