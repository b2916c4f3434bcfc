"""Debug: ray counts of vrt_render_device_noalloc replayed from a captured graph."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
import raytracing_amd as ra
from raytracing_amd import abi, host
ctx = host.Context(0)
ctx.set_scene(ra.random_spheres_scene(400, seed=4))
W, H = 1024, 512
ctx.set_camera(ra.random_spheres_camera(W, H))
rgba = torch.zeros((H, W, 4), device="cuda"); hit = torch.zeros((H, W), dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
t = host.sampler_table(abi.VRT_SAMPLER_JITTERED, 4, 5, 0.01, 0.01, seed=101)
prm = ra.Params(spp=4, offsets=t, seed=5, max_depth=4, rng=200.0)
for k in range(2):
    ctx.render_device_noalloc(prm, rgba.data_ptr(), hit.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    print("plain", k, list(ctx.fetch_stats().rays))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    ctx.render_device_noalloc(prm, rgba.data_ptr(), hit.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("after capture", list(ctx.fetch_stats().rays))
for k in range(3):
    g.replay(); torch.cuda.synchronize()
    print("replay", k, list(ctx.fetch_stats().rays))
ctx.render_device_noalloc(prm, rgba.data_ptr(), hit.data_ptr(), side.cuda_stream)
torch.cuda.synchronize()
print("plain again", list(ctx.fetch_stats().rays))
