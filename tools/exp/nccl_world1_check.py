"""One-rank RCCL group on one GPU: the device-tensor code paths of tiles.broadcast_scene and
FrameGather's collective calls (gather with async_op, wait on a side stream) run without error."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.distributed as dist
import raytracing_amd as ra
from raytracing_amd import tiles
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.cuda.set_device(0)
for scene in (ra.teaching_scene(), ra.textured_quad_scene() if hasattr(ra, "textured_quad_scene") else ra.teaching_scene()):
    got = tiles.broadcast_scene(scene, dev)
    assert np.array_equal(got.spheres, scene.spheres) and np.array_equal(got.boxes, scene.boxes)
fg = tiles.FrameGather(64, 32, 3, 2, 0, 8, dev)      # built as rank 0 of 2: buffers and index as in a 2-rank run
outs = [fg.gathered[0]]                               # but gathered over the 1-rank group
side = torch.cuda.Stream(device=dev)
fg.local.fill_(1.0)
h = dist.gather(fg.local, outs, dst=0, async_op=True)
with torch.cuda.stream(side):
    h.wait()
    ev = torch.cuda.Event(); ev.record(side)
    fg.deinterleave()
torch.cuda.current_stream().wait_event(ev)
torch.cuda.synchronize()
rows = tiles.owned_rows(64, 0, 2, 8)
assert bool((fg.final[rows] == 1.0).all())
print("nccl world-1 check ok")
dist.destroy_process_group()
