"""Study: megakernel tiles dispatched most expensive first, costs taken from a first frame (clock ticks per tile).
Run on the GPU box: python3 tools/exp_tile_order.py [workload] [steps]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
import raytracing_amd as ra
from raytracing_amd import host
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
scene, cam, offsets, kw, label = bench.build_workload(wl, 1)
ctx = host.Context(0); ctx.set_scene(scene); ctx.set_camera(cam)
prm = ra.Params(offsets=offsets, **kw)
H, W = cam.height, cam.width
dev = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream()
def frame_ms(n):
    ctx.render_device(prm, dev.data_ptr(), None, s.cuda_stream); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(n): ctx.render_device(prm, dev.data_ptr(), None, s.cuda_stream)
    e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
base = frame_ms(steps)
ref = dev.clone()
n_tiles = 1 << 26   # more than any frame here needs
cost = torch.zeros(n_tiles, dtype=torch.int32, device="cuda")
ctx.set_option("tile_cost_ptr", cost.data_ptr())
with_cost = frame_ms(steps)
nz = int((cost != 0).sum()); n_blocks = int(torch.nonzero(cost)[-1]) + 1
c = cost[:n_blocks].to(torch.int64) & 0xffffffff
print(wl, "tiles", n_blocks, "cost min/median/mean/max", int(c.min()), int(c.median()), float(c.float().mean()), int(c.max()))
order = torch.argsort(c, descending=True, stable=True).to(torch.int32).contiguous()
ctx.set_option("tile_order_ptr", order.data_ptr())
sorted_ms = frame_ms(steps)
print("equal", bool(torch.equal(torch.nan_to_num(dev), torch.nan_to_num(ref))))
ctx.set_option("tile_cost_ptr", 0)
sorted_nocost = frame_ms(steps)
rnd = torch.randperm(n_blocks, device="cuda").to(torch.int32).contiguous()
ctx.set_option("tile_order_ptr", rnd.data_ptr())
random_ms = frame_ms(steps)
ctx.set_option("tile_order_ptr", 0)
again = frame_ms(steps)
print(f"{wl}: as shipped {base:.4f} ms, recording costs {with_cost:.4f}, costly tiles first {sorted_ms:.4f} (without recording {sorted_nocost:.4f}), random order {random_ms:.4f}, as shipped again {again:.4f}")
