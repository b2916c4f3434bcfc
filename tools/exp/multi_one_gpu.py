"""What the multi-GPU context costs on the host side, measured on ONE GPU named n times
(vrt_create_multi with devices = [0] * n): the n strips sets share one card and one PCIe link, so
the frame cannot get faster; what shows is the price of the split itself (one host thread per
sub-context, n launches, n deliveries) against the one-GPU context, for per-GPU delivery and for
strips gathered on the first GPU.  usage: python tools/exp/multi_one_gpu.py [workload] [n]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import raytracing_amd as ra  # noqa: E402
from raytracing_amd import host  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = 200 if wl == "c2" else 5
scene, cam, offsets, kw, label = bench.build_workload(wl, 1)
prm = ra.Params(offsets=offsets, **kw)


def timed(ctx, rgba, hit):
    ctx.render(prm, rgba=rgba, hit=hit)
    ctx.render(prm, rgba=rgba, hit=hit)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        ctx.render(prm, rgba=rgba, hit=hit)
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


one = host.Context(0)
one.set_scene(scene); one.set_camera(cam)
want, want_hit, _ = one.render(prm)
r1, h1 = one.host_alloc(want.shape, np.float32), one.host_alloc(want_hit.shape, np.int32)
res = {"workload": label, "one_gpu_context_ms": timed(one, r1, h1)}
one.close()
m = host.Context(devices=[0] * n)
m.set_scene(scene); m.set_camera(cam)
rgba, hit = m.host_alloc(want.shape, np.float32), m.host_alloc(want_hit.shape, np.int32)
for gather in (0, 1):
    m.set_option("multi_gather", gather)
    rgba[:] = 0; hit[:] = -9
    ms = timed(m, rgba, hit)
    same = bool(np.array_equal(rgba, want, equal_nan=True) and np.array_equal(hit, want_hit))
    res["gathered_ms" if gather else "per_gpu_delivery_ms"] = ms
    res["gathered_same_bits" if gather else "per_gpu_delivery_same_bits"] = same
m.close()
res["sub_contexts_on_gpu_0"] = n
print(json.dumps(res))
