"""vrt_render_device_noalloc (the entry point for callers that cannot allocate device memory)
must deliver what vrt_render_device delivers, bit for bit, call after call: its counters and its
copy of the sampler table live in a static device array that keeps the previous call's contents."""
import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import abi, host

pytestmark = pytest.mark.gpu


def _both(ctx, torch, prm, H, W):
    """The frame, hit ids and ray counts of vrt_render_device and of the no-alloc entry point."""
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for call in (ctx.render_device, ctx.render_device_noalloc):
        rgba = torch.full((H, W, 4), -3.0, device="cuda")
        hit = torch.full((H, W), -9, dtype=torch.int32, device="cuda")
        call(prm, rgba.data_ptr(), hit.data_ptr(), st)
        torch.cuda.synchronize()
        rays = list(ctx.fetch_stats().rays)
        out.append((rgba.cpu().numpy(), hit.cpu().numpy(), rays))
    return out


@pytest.mark.parametrize("bvh", [False, True])
def test_noalloc_equals_render_device_twice_second_smaller(vrt_ctx, bvh):
    import torch
    scene = ra.random_spheres_scene(400, seed=4) if bvh else ra.teaching_scene()
    vrt_ctx.set_scene(scene)
    rng = 200.0 if bvh else 30.0
    # first call: 16 samples per pixel from 9 sets on a 160x96 frame;
    # second call: a smaller frame, fewer samples, a smaller table (the array still holds the first one's)
    for (W, H, spp, sets, depth) in ((160, 96, 16, 9, 6), (75, 53, 3, 2, 4)):
        cam = ra.random_spheres_camera(W, H) if bvh else ra.teaching_camera(W, H)
        vrt_ctx.set_camera(cam)
        offs = host.sampler_table(abi.VRT_SAMPLER_JITTERED if spp == 16 else abi.VRT_SAMPLER_RANDOM,
                                  spp, sets, 0.02, 0.02, seed=spp)
        prm = ra.Params(spp=spp, offsets=offs, seed=5, max_depth=depth, rng=rng)
        (a_rgba, a_hit, a_rays), (b_rgba, b_hit, b_rays) = _both(vrt_ctx, torch, prm, H, W)
        assert sum(a_rays) > 0 and (a_hit >= 0).any()
        assert np.array_equal(a_rgba, b_rgba, equal_nan=True), "frame differs"
        assert np.array_equal(a_hit, b_hit), "hit ids differ"
        assert a_rays == b_rays, "ray counts differ"
    # without a table (1 sample per pixel) and in the strip format of a multi-GPU split
    cam = ra.random_spheres_camera(64, 40) if bvh else ra.teaching_camera(64, 40)
    vrt_ctx.set_camera(cam)
    prm = ra.Params(spp=1, max_depth=3, rng=rng, tile_index=1, tile_count=2, tile_rows=8, compact_rows=True,
                    flags=abi.VRT_OUT_LINEAR_RGB)
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for call in (vrt_ctx.render_device, vrt_ctx.render_device_noalloc):
        rgb = torch.zeros((16, 64, 3), device="cuda")
        call(prm, rgb.data_ptr(), None, st)
        torch.cuda.synchronize()
        res.append((rgb.cpu().numpy(), list(vrt_ctx.fetch_stats().rays)))
    assert np.array_equal(res[0][0], res[1][0], equal_nan=True) and res[0][1] == res[1][1]


def test_noalloc_rejects_what_does_not_fit(vrt_ctx):
    import torch
    vrt_ctx.set_scene(ra.teaching_scene())
    vrt_ctx.set_camera(ra.teaching_camera(32, 32))
    buf = torch.zeros((32, 32, 4), device="cuda")
    spp, sets = 256, abi.VRT_NOALLOC_MAX_OFFSETS // 256 + 1
    big = np.zeros((sets, spp, 2), np.float32)
    with pytest.raises(host.VrtError) as e:
        vrt_ctx.render_device_noalloc(ra.Params(spp=spp, offsets=big), buf.data_ptr())
    assert e.value.code == abi.VRT_ERR_CAPACITY and "VRT_NOALLOC_MAX_OFFSETS" in str(e.value)
    with pytest.raises(host.VrtError) as e:
        vrt_ctx.render_device_noalloc(ra.Params(spp=1, mode=abi.VRT_MODE_WAVEFRONT), buf.data_ptr())
    assert e.value.code == abi.VRT_ERR_INVALID
    # the largest table that fits is accepted
    ok = np.zeros((abi.VRT_NOALLOC_MAX_OFFSETS // 256, 256, 2), np.float32)
    vrt_ctx.render_device_noalloc(ra.Params(spp=256, offsets=ok, max_depth=1), buf.data_ptr())
    torch.cuda.synchronize()


def test_noalloc_leaves_device_memory_alone_and_can_be_captured(vrt_ctx):
    """What the entry point is named for: a call on a layout large enough for the tile-order
    machinery of the other entry points (32 768 tiles), with a sampler table the GPU has not seen,
    changes nothing in the device's free memory.  Then the use it is documented for: with the table
    in place, the same call under stream capture becomes a graph whose replay renders the frame
    (a new table under capture is refused, not silently uploaded)."""
    import torch
    vrt_ctx.set_scene(ra.random_spheres_scene(400, seed=4))
    W, H = 1024, 512
    vrt_ctx.set_camera(ra.random_spheres_camera(W, H))
    rgba = torch.zeros((H, W, 4), device="cuda")
    hit = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    tables = [host.sampler_table(abi.VRT_SAMPLER_JITTERED, 4, 5, 0.01, 0.01, seed=s) for s in (101, 102, 103)]
    prm = [ra.Params(spp=4, offsets=t, seed=5, max_depth=4, rng=200.0) for t in tables]
    vrt_ctx.render_device_noalloc(prm[0], rgba.data_ptr(), hit.data_ptr(), side.cuda_stream)   # warm: code objects, streams
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for k in (1, 2, 1):
        vrt_ctx.render_device_noalloc(prm[k], rgba.data_ptr(), hit.data_ptr(), side.cuda_stream)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0, "vrt_render_device_noalloc changed the device's free memory"
    want, want_hit, want_rays = rgba.cpu().numpy(), hit.cpu().numpy(), list(vrt_ctx.fetch_stats().rays)
    ref = torch.zeros_like(rgba)
    vrt_ctx.render_device(prm[1], ref.data_ptr(), None, side.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(ref.cpu().numpy(), want, equal_nan=True)
    # under capture: the table of the last call is in place
    graph = torch.cuda.CUDAGraph()
    rgba.zero_(); hit.fill_(-9)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        vrt_ctx.render_device_noalloc(prm[1], rgba.data_ptr(), hit.data_ptr(), torch.cuda.current_stream().cuda_stream)
        with pytest.raises(host.VrtError) as e:   # a table the array does not hold cannot be uploaded here
            vrt_ctx.render_device_noalloc(prm[0], rgba.data_ptr(), hit.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert e.value.code == abi.VRT_ERR_STATE
    torch.cuda.synchronize()
    assert (hit.cpu().numpy() == -9).all()   # captured, not run
    for _ in range(2):
        rgba.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert np.array_equal(rgba.cpu().numpy(), want, equal_nan=True) and np.array_equal(hit.cpu().numpy(), want_hit)
        assert list(vrt_ctx.fetch_stats().rays) == want_rays
