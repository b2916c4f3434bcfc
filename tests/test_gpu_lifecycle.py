"""Contexts can be created and destroyed over and over, scenes replaced and modes switched
without device memory piling up (the Unity host re-marshals the scene on every edit and creates
a context per play session)."""
import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import abi, host

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def test_create_render_destroy_does_not_leak():
    import torch
    torch.zeros(1, device="cuda")   # the runtime's own pools exist before the first reading
    scenes = [(ra.teaching_scene(), ra.teaching_camera(96, 64), 30.0),
              (ra.random_spheres_scene(n=600, seed=5), ra.random_spheres_camera(96, 54), 200.0)]
    frames = {}
    readings = []
    for it in range(12):
        ctx = host.Context(0)
        for k, (scene, cam, rng) in enumerate(scenes):
            ctx.set_scene(scene)
            ctx.set_camera(cam)
            for mode in (abi.VRT_MODE_MEGAKERNEL, abi.VRT_MODE_WAVEFRONT):
                rgba, hit, st = ctx.render(ra.Params(spp=2, max_depth=4, rng=rng, mode=mode))
                key = (k, mode)
                if key in frames:   # and every round gives the same frame
                    assert np.array_equal(frames[key], rgba, equal_nan=True)
                else:
                    frames[key] = rgba
        ctx.close()
        readings.append(_free_bytes())
    # the first rounds may grow allocator pools; after that nothing may be lost per round
    assert readings[-1] >= readings[3] - (8 << 20), readings


def test_scene_replacement_in_one_context_does_not_leak(vrt_ctx):
    cam = ra.random_spheres_camera(64, 36)
    vrt_ctx.set_camera(cam)
    readings = []
    for it in range(10):
        vrt_ctx.set_scene(ra.random_spheres_scene(n=300 + 50 * (it % 3), seed=it))
        vrt_ctx.render(ra.Params(spp=1, max_depth=3, rng=200.0))
        readings.append(_free_bytes())
    assert readings[-1] >= readings[3] - (8 << 20), readings
