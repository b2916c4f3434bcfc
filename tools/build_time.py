"""Tree build time per workload: device time between the build's events and wall time of vrt_set_scene, five builds in a row
(the first includes module load and first allocations).  Run on the GPU box: python3 tools/build_time.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import bench
import raytracing_amd as ra
from raytracing_amd import host
for wl in ("c4", "c3"):
    scene, cam, offsets, kw, label = bench.build_workload(wl, 1)
    ctx = host.Context(0)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.set_scene(scene); ts.append((time.perf_counter() - t0) * 1e3)
        b = ctx.fetch_stats().bvh_build_ms
    print(wl, "build_ms(device events)", round(b, 2), "set_scene wall ms", [round(t, 1) for t in ts])
    ctx.close()
