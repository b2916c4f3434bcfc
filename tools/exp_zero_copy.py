import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
import raytracing_amd as ra
from raytracing_amd import host
scene, cam, offsets, kw, label = bench.build_workload("c2", 1)
ctx = host.Context(0); ctx.set_scene(scene); ctx.set_camera(cam)
prm = ra.Params(offsets=offsets, **kw)
H, W = cam.height, cam.width
pin = torch.empty((H, W, 4), dtype=torch.float32, pin_memory=True)
pin_hit = torch.empty((H, W), dtype=torch.int32, pin_memory=True)
dev = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream()
def t(fn, n=30):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
print("device frame      ", t(lambda: ctx.render_device(prm, dev.data_ptr(), None, s.cuda_stream)))
print("zero-copy rgba    ", t(lambda: ctx.render_device(prm, pin.data_ptr(), None, s.cuda_stream)))
print("zero-copy rgba+hit", t(lambda: ctx.render_device(prm, pin.data_ptr(), pin_hit.data_ptr(), s.cuda_stream)))
ref, ref_hit, _ = ctx.render(prm)
print("equal", np.array_equal(pin.numpy(), ref, equal_nan=True), np.array_equal(pin_hit.numpy(), ref_hit))
p_rgba, p_hit = ctx.host_alloc((H, W, 4), np.float32), ctx.host_alloc((H, W), np.int32)
print("vrt_render pinned ", t(lambda: ctx.render(prm, rgba=p_rgba, hit=p_hit)))
print("vrt_render pinned no hit", t(lambda: ctx.render(prm, rgba=p_rgba, hit=None, want_hit=False)))
