"""vrt_render of the 1080p teaching-scene frame as Color32 into page-locked memory: stored by the
kernel (zero_copy 1) against rendered into device memory and copied (zero_copy 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import raytracing_amd as ra
from raytracing_amd import abi, host
import ctypes as C
ctx = host.Context(0)
ctx.set_scene(ra.teaching_scene()); ctx.set_camera(ra.teaching_camera(1920, 1080))
buf8 = ctx.host_alloc((1080, 1920, 4), np.uint8)
buf32 = ctx.host_alloc((1080, 1920, 4), np.float32)
for flags, buf, name in ((abi.VRT_OUT_RGBA8, buf8, "rgba8"), (0, buf32, "float")):
    prm = ra.Params(spp=1, max_depth=5, flags=flags)
    for zc in (1, 0):
        ctx.set_option("zero_copy", zc)
        for stats in (True, False):
            st = abi.vrt_stats()
            def call():
                rc = ctx._lib.vrt_render(ctx._h, C.byref(prm.c), buf.ctypes.data, None, C.byref(st) if stats else None)
                assert rc == 0
            for _ in range(5): call()
            ts = []
            for _ in range(60):
                t0 = time.perf_counter(); call(); ts.append((time.perf_counter() - t0) * 1e3)
            print(f"{name} zero_copy={zc} stats={stats}: {np.median(ts):.4f} ms (min {min(ts):.4f})")
