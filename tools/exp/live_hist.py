"""Study: turns of the megakernel's bounce loop by the number of lanes still on a path
(library built with -DVRT_STUDY_LIVE_HIST=1, selected with VRT_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import raytracing_amd as ra
from raytracing_amd import host
for wl in sys.argv[1:]:
    scene, cam, offsets, kw, label = bench.build_workload(wl, 1)
    ctx = host.Context(0)
    ctx.set_scene(scene); ctx.set_camera(cam)
    buf = ctx.host_alloc((cam.height, cam.width, 4), np.float32)
    _, _, st = ctx.render(ra.Params(offsets=offsets, **kw), rgba=buf, want_hit=False)
    h = list(st.rays) + [st.nodes_visited, st.prims_tested]
    tot = float(sum(h))
    print(label)
    for name, v in zip(("1", "2", "3-4", "5-8", "9-16", "17-32", "33-48", "49-64"), h):
        print(f"  {name:>6} lanes: {v:12d} turns  {100 * v / tot:5.1f} %")
    ctx.close()
