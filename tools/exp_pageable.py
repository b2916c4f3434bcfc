"""vrt_render into pageable numpy arrays: staged bands (option staged=1) against the runtime's pageable copy (staged=0).
Run on the GPU box: python3 tools/exp_pageable.py [workload]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import bench
import raytracing_amd as ra
from raytracing_amd import host
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
scene, cam, offsets, kw, label = bench.build_workload(wl, 1)
ctx = host.Context(0); ctx.set_scene(scene); ctx.set_camera(cam)
prm = ra.Params(offsets=offsets, **kw)
H, W = cam.height, cam.width
def t(fn, n=20):
    fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
a = np.zeros((H, W, 4), np.float32); h = np.zeros((H, W), np.int32)
for st in (1, 0):
    ctx.set_option("staged", st)
    a[:] = 0; h[:] = 0
    _, _, s1 = ctx.render(prm, rgba=a, hit=h)
    if st: ref, refh, rays = a.copy(), h.copy(), list(s1.rays)
    else: print("equal", np.array_equal(a, ref, equal_nan=True), np.array_equal(h, refh), list(s1.rays) == rays, s1.pixels)
    print("staged", st, "pageable rgba+hit", round(t(lambda: ctx.render(prm, rgba=a, hit=h)), 3), "rgba only", round(t(lambda: ctx.render(prm, rgba=a, hit=None, want_hit=False)), 3))
