"""Host-side cost of one pipelined frame step (render launch + async gather + wait + de-interleave)
with a one-rank RCCL group on one GPU: an estimate of what rank 0's Python thread spends per frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
x = torch.zeros((136, 1920, 4), device="cuda")
outs = torch.zeros((1, 136, 1920, 4), device="cuda")
final = torch.zeros((137, 1920, 4), device="cuda")
idx = torch.arange(136, device="cuda")
for n in (20, 2000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        h = dist.gather(x, list(outs.unbind(0)), dst=0, async_op=True)
        h.wait()
        final.index_copy_(0, idx, outs.view(136, 1920, 4))
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(n, "host us/step", (t1 - t0) / n * 1e6, "incl. drain", (t2 - t0) / n * 1e6)
dist.destroy_process_group()
