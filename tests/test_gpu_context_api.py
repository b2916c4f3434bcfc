"""Context-level behaviour of the C ABI on a GPU: the multi-GPU context (vrt_create_multi), frames
delivered in parts, transactional scene replacement, page-locked caller buffers, the sampler-table
cache under frames in flight, and error reporting of the wavefront queues."""
import time

import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import abi, host
from conftest import assert_frame_matches_oracle

pytestmark = pytest.mark.gpu
_ORACLE = {}


def _oracle_frame(oracle_mod, key, scene, cam, kw):
    """Oracle frames are rendered once per session and scene."""
    if key not in _ORACLE:
        _ORACLE[key] = oracle_mod.render(scene, cam, ra.Params(**kw).c, n_threads=8)
    return _ORACLE[key]


def _spheres(w=160, h=93):
    offs = host.sampler_table(abi.VRT_SAMPLER_JITTERED, 4, 7, 0.01, 0.01, seed=2)
    return (ra.random_spheres_scene(1200, seed=9), ra.random_spheres_camera(w, h),
            dict(spp=4, offsets=offs, seed=3, max_depth=5, rng=200.0))


@pytest.mark.parametrize("gather", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 3])
@pytest.mark.parametrize("which", ["teaching", "spheres"])
def test_multi_gpu_context_equals_one_gpu(vrt_ctx, oracle_mod, which, n, gather):
    """vrt_create_multi renders strips on every GPU of the context; every GPU delivers its strips
    to the caller's buffer itself (gather = 0), or the strips are gathered on the first GPU and
    tone-mapped there (option multi_gather); the frame, the hit ids and the ray counts are the
    oracle's and, bit for bit, those of the one-GPU context.  A one-GPU box stands in for the node
    by naming GPU 0 several times (gathered strips then travel by peer-to-peer copies): strip
    bookkeeping, ragged last strips, assembly, statistics, path records, frames into page-locked
    buffers of the context and the non-blocking calls all run as they do across GPUs."""
    if which == "teaching":
        scene, cam, kw = ra.teaching_scene(), ra.teaching_camera(150, 85), dict(spp=1, max_depth=5)
    else:
        scene, cam, kw = _spheres()
    vrt_ctx.set_scene(scene)
    vrt_ctx.set_camera(cam)
    want, want_hit, want_st = vrt_ctx.render(ra.Params(**kw))
    oracle_frame = _oracle_frame(oracle_mod, which, scene, cam, kw)
    m = host.Context(devices=[0] * n)
    try:
        m.set_option("multi_gather", gather)
        m.set_scene(scene)
        m.set_camera(cam)
        for rows in (8, 5):
            got, got_hit, st = m.render(ra.Params(tile_rows=rows, **kw))
            assert_frame_matches_oracle(got, got_hit, st.rays, oracle_frame)
            assert np.array_equal(got, want, equal_nan=True) and np.array_equal(got_hit, want_hit)
            assert list(st.rays) == list(want_st.rays) and st.pixels == cam.width * cam.height
            assert st.gpu_ms > 0
        # without hit ids, and with path records (traced on the first GPU)
        got, none, _ = m.render(ra.Params(**kw), want_hit=False)
        assert none is None and np.array_equal(got, want, equal_nan=True)
        prm = ra.Params(**kw)
        px = [3 * cam.width + 7, 40 * cam.width + 90]
        segs, counts = prm.record(px, 64)
        m.render(prm)
        ref = ra.Params(**kw)
        rsegs, rcounts = ref.record(px, 64)
        vrt_ctx.render(ref)
        assert list(counts) == list(rcounts) and min(counts) > 0
        for k, pix in enumerate(px):   # the recorded ray trees are the oracle's
            _, _, o_segs, o_n, _ = oracle_mod.trace_pixel(scene, cam, ra.Params(**kw).c, pix // cam.width, pix % cam.width, -1, 64)
            assert counts[k] == o_n
            for j in range(min(o_n, 64)):
                a, b = segs[k * 64 + j], o_segs[j]
                assert (a.kind, a.depth, a.hit_prim, a.sample) == (b.kind, b.depth, b.hit_prim, b.sample)
                assert list(a.origin) == list(b.origin) and list(a.end) == list(b.end)
        for k, n_seg in enumerate(counts):   # entries beyond a pixel's count are not written
            assert bytes(segs)[k * 64 * 40:(k * 64 + n_seg) * 40] == bytes(rsegs)[k * 64 * 40:(k * 64 + n_seg) * 40]
        # a frame into a page-locked buffer of the context: every GPU's kernel stores its strips there
        # itself; the rest of the allocation keeps what the caller had there
        big = m.host_alloc((cam.height + 6, cam.width, 4), np.float32)
        big[:] = -3.0
        out = big[3:3 + cam.height]
        _, got_hit, st = m.render(ra.Params(tile_rows=5, **kw), rgba=out)
        assert np.array_equal(out, want, equal_nan=True) and np.array_equal(got_hit, want_hit)
        assert (big[:3] == -3.0).all() and (big[3 + cam.height:] == -3.0).all() and list(st.rays) == list(want_st.rays)
        # the non-blocking calls: every GPU's part is in flight at once, the frame is done when all are
        big[:] = -3.0
        hit2 = np.full_like(want_hit, -7)
        m.render_begin(ra.Params(**kw), out, hit2)
        with pytest.raises(host.VrtError) as e:
            m.render(ra.Params(**kw))
        assert e.value.code == abi.VRT_ERR_STATE
        deadline = time.time() + 60
        while not m.render_poll():
            assert time.time() < deadline
        st = m.render_end()
        assert_frame_matches_oracle(out, hit2, st.rays, oracle_frame)
        assert np.array_equal(out, want, equal_nan=True) and np.array_equal(hit2, want_hit)
        assert st.pixels == cam.width * cam.height and st.gpu_ms > 0
        with pytest.raises(host.VrtError) as e:
            m.render_end()
        assert e.value.code == abi.VRT_ERR_STATE
        m.host_free(big)
        # the calls that make no sense across GPUs say so
        with pytest.raises(host.VrtError) as e:
            m.render_begin_parts(ra.Params(**kw), got, None, 2)
        assert e.value.code == abi.VRT_ERR_INVALID
        with pytest.raises(host.VrtError):
            m.render(ra.Params(tile_index=0, tile_count=2, **kw))
        # per-pixel and per-ray calls run on the first GPU
        a = m.trace_pixel(ra.Params(**kw), 11, 13)
        b = vrt_ctx.trace_pixel(ra.Params(**kw), 11, 13)
        assert np.array_equal(a[0], b[0], equal_nan=True) and a[1] == b[1] and a[3] == b[3]
    finally:
        m.close()


def test_multi_gpu_scene_replacement_is_all_or_nothing(vrt_ctx):
    scene, cam, kw = _spheres(96, 64)
    m = host.Context(devices=[0, 0])
    try:
        m.set_scene(scene)
        m.set_camera(cam)
        before, _, _ = m.render(ra.Params(**kw))
        m.set_option("fail_after_build", 1)
        with pytest.raises(host.VrtError):
            m.set_scene(ra.random_spheres_scene(500, seed=1))
        m.set_option("fail_after_build", 0)
        after, _, _ = m.render(ra.Params(**kw))
        assert np.array_equal(before, after, equal_nan=True)
    finally:
        m.close()


def test_failed_scene_upload_keeps_the_old_scene(vrt_ctx, vrt_opts):
    """A failure after validation (uploads, tree build) must leave the context on its old scene:
    the new one is assembled beside it and swapped in only on success."""
    scene, cam, kw = _spheres(96, 64)
    vrt_ctx.set_scene(scene)
    vrt_ctx.set_camera(cam)
    before, bh, bst = vrt_ctx.render(ra.Params(**kw))
    vrt_opts("fail_after_build", 1)
    with pytest.raises(host.VrtError) as e:
        vrt_ctx.set_scene(ra.tessellated_mesh_scene(quads=20))
    assert e.value.code == abi.VRT_ERR_HIP
    vrt_opts("fail_after_build", 0)
    after, ah, ast = vrt_ctx.render(ra.Params(**kw))
    assert np.array_equal(before, after, equal_nan=True) and np.array_equal(bh, ah)
    assert list(bst.rays) == list(ast.rays) and ast.bvh_nodes == bst.bvh_nodes


@pytest.mark.parametrize("n_parts", [1, 3, 7])
def test_frame_delivered_in_parts(vrt_ctx, oracle_mod, n_parts):
    """vrt_render_begin_parts delivers the frame strip set by strip set; a part's rows are final
    once vrt_render_poll_parts counts it, the other rows of the caller's buffer are untouched
    until their part lands, and the statistics are those of the whole frame."""
    scene, cam, kw = _spheres(200, 117)
    vrt_ctx.set_scene(scene)
    vrt_ctx.set_camera(cam)
    ref, ref_hit, ref_st = vrt_ctx.render(ra.Params(**kw))
    rgba = vrt_ctx.host_alloc(ref.shape, np.float32)
    rgba[:] = -7.0
    hit = np.full_like(ref_hit, -7)
    try:
        vrt_ctx.render_begin_parts(ra.Params(tile_rows=8, **kw), rgba, hit, n_parts)
        seen, deadline = 0, time.time() + 60
        while seen < n_parts:
            done = vrt_ctx.render_poll_parts()
            assert done >= seen and time.time() < deadline
            for k in range(seen, done):
                rows = [h for h in range(cam.height) if (h // 8) % n_parts == k]
                assert np.array_equal(rgba[rows], ref[rows], equal_nan=True)
                assert np.array_equal(hit[rows], ref_hit[rows])
            seen = done
        assert vrt_ctx.render_poll()
        st = vrt_ctx.render_end()
        assert_frame_matches_oracle(rgba, hit, st.rays, _oracle_frame(oracle_mod, "parts", scene, cam, kw))
        assert np.array_equal(rgba, ref, equal_nan=True) and np.array_equal(hit, ref_hit)
        assert list(st.rays) == list(ref_st.rays) and st.pixels == cam.width * cam.height
        with pytest.raises(host.VrtError) as e:
            vrt_ctx.render_poll_parts()
        assert e.value.code == abi.VRT_ERR_STATE
        with pytest.raises(host.VrtError) as e:
            vrt_ctx.render_begin_parts(ra.Params(**kw), rgba, hit, 0)
        assert e.value.code == abi.VRT_ERR_INVALID
    finally:
        vrt_ctx.host_free(rgba)


def test_strip_calls_copy_only_their_rows(vrt_ctx):
    """A strip-sharded call into a whole-frame buffer writes the rows it owns and nothing else."""
    scene, cam, kw = _spheres(120, 70)
    vrt_ctx.set_scene(scene)
    vrt_ctx.set_camera(cam)
    ref, ref_hit, _ = vrt_ctx.render(ra.Params(**kw))
    rgba = np.full_like(ref, -3.0)
    hit = np.full_like(ref_hit, -3)
    vrt_ctx.render(ra.Params(tile_index=1, tile_count=3, tile_rows=8, **kw), rgba=rgba, hit=hit)
    mine = np.array([(h // 8) % 3 == 1 for h in range(cam.height)])
    assert np.array_equal(rgba[mine], ref[mine], equal_nan=True) and np.array_equal(hit[mine], ref_hit[mine])
    assert (rgba[~mine] == -3.0).all() and (hit[~mine] == -3).all()


def test_page_locked_host_buffers(vrt_ctx):
    """Frames land in buffers of vrt_host_alloc like in any other; registering a caller's own buffer
    is allowed to be refused by the operating system, and then the buffer still works."""
    scene, cam, kw = _spheres(96, 64)
    vrt_ctx.set_scene(scene)
    vrt_ctx.set_camera(cam)
    ref, ref_hit, _ = vrt_ctx.render(ra.Params(**kw))
    rgba, hit = vrt_ctx.host_alloc(ref.shape, np.float32), vrt_ctx.host_alloc(ref_hit.shape, np.int32)
    rgba[:] = 0
    vrt_ctx.render(ra.Params(**kw), rgba=rgba, hit=hit)
    assert np.array_equal(rgba, ref, equal_nan=True) and np.array_equal(hit, ref_hit)
    vrt_ctx.host_free(hit)
    vrt_ctx.host_free(rgba)
    with pytest.raises(host.VrtError) as e:
        vrt_ctx.host_free(ref)
    assert e.value.code == abi.VRT_ERR_INVALID
    own = np.zeros_like(ref)
    try:
        vrt_ctx.host_register(own)
        registered = True
    except host.VrtError as e:
        assert e.code == abi.VRT_ERR_HIP
        registered = False
    vrt_ctx.render(ra.Params(**kw), rgba=own, hit=None, want_hit=False)
    assert np.array_equal(own, ref, equal_nan=True)
    if registered:
        with pytest.raises(host.VrtError) as e:
            vrt_ctx.host_register(own)
        assert e.value.code == abi.VRT_ERR_STATE
        vrt_ctx.host_unregister(own)
    with pytest.raises(host.VrtError) as e:
        vrt_ctx.host_unregister(own)
    assert e.value.code == abi.VRT_ERR_INVALID


def test_frames_in_flight_keep_their_sampler_table(vrt_ctx):
    """Frames queued on a caller stream with different sampler tables (more tables than the cache
    has slots) must each sample with their own table, whatever is uploaded after them."""
    import torch
    scene, cam = ra.teaching_scene(), ra.teaching_camera(320, 180)
    vrt_ctx.set_scene(scene)
    vrt_ctx.set_camera(cam)
    tables = [host.sampler_table(abi.VRT_SAMPLER_RANDOM, 4, 3, 0.02, 0.02, seed=s) for s in range(6)]
    prms = [ra.Params(spp=4, offsets=t, seed=1, max_depth=5) for t in tables]
    want = [vrt_ctx.render(p)[0].copy() for p in prms]
    side = torch.cuda.Stream()
    outs = [torch.empty((180, 320, 4), device="cuda") for _ in prms]
    for _ in range(3):
        for p, o in zip(prms, outs):
            vrt_ctx.render_device(p, o.data_ptr(), None, side.cuda_stream)
    torch.cuda.synchronize()
    for w, o in zip(want, outs):
        assert np.array_equal(o.cpu().numpy(), w, equal_nan=True)


def test_wavefront_overflow_is_reported_without_stats(vrt_ctx, vrt_opts):
    """A wavefront frame whose ray queues overflowed has dropped paths; every blocking call must
    say so whether or not the caller asked for statistics (the queues are cut short here)."""
    import ctypes as C
    scene, cam = ra.teaching_scene(), ra.teaching_camera(64, 48)
    vrt_ctx.set_scene(scene)
    vrt_ctx.set_camera(cam)
    kw = dict(spp=1, max_depth=6, mode=abi.VRT_MODE_WAVEFRONT)
    ref, _, ref_st = vrt_ctx.render(ra.Params(**kw))
    assert ref_st.rays[abi.VRT_RAY_REFLECT] > 400
    vrt_opts("wf_queue_cap", 300)
    rgba = np.zeros_like(ref)
    prm = ra.Params(**kw)
    with pytest.raises(host.VrtError) as e:
        vrt_ctx.render(prm)
    assert e.value.code == abi.VRT_ERR_CAPACITY
    with pytest.raises(host.VrtError) as e:   # no statistics asked for
        vrt_ctx._chk(vrt_ctx._lib.vrt_render(vrt_ctx._h, C.byref(prm.c), rgba.ctypes.data, None, None))
    assert e.value.code == abi.VRT_ERR_CAPACITY
    vrt_ctx.render_begin(prm, rgba)
    with pytest.raises(host.VrtError) as e:
        vrt_ctx._chk(vrt_ctx._lib.vrt_render_end(vrt_ctx._h, None))
    assert e.value.code == abi.VRT_ERR_CAPACITY
    vrt_opts("wf_queue_cap", 0)
    again, _, st = vrt_ctx.render(prm)
    assert np.array_equal(again, ref, equal_nan=True) and list(st.rays) == list(ref_st.rays)


def test_frames_stored_straight_into_page_locked_buffers(vrt_ctx):
    """A frame buffer from vrt_host_alloc is written by the frame kernels themselves instead of
    being copied after them.  Same bits as with the copy (option zero_copy off) for whole frames in
    both kernel organisations, for one strip set of three (the other rows keep what the caller had
    there), for packed rows, and for a frame that starts part-way into the allocation."""
    scene, cam, kw = _spheres(96, 64)
    vrt_ctx.set_scene(scene)
    vrt_ctx.set_camera(cam)
    ref, ref_hit, ref_st = vrt_ctx.render(ra.Params(**kw))
    big = vrt_ctx.host_alloc((2 * cam.height, cam.width, 4), np.float32)
    try:
        for mode in (abi.VRT_MODE_MEGAKERNEL, abi.VRT_MODE_WAVEFRONT):
            for zc in (1, 0):
                vrt_ctx.set_option("zero_copy", zc)
                big[:] = -3.0
                out = big[5:5 + cam.height]          # not at the start of the allocation
                _, hit, st = vrt_ctx.render(ra.Params(mode=mode, **kw), rgba=out)
                assert np.array_equal(out, ref, equal_nan=True) and np.array_equal(hit, ref_hit)
                assert list(st.rays) == list(ref_st.rays)
                assert (big[:5] == -3.0).all() and (big[5 + cam.height:] == -3.0).all()
        for zc in (1, 0):
            vrt_ctx.set_option("zero_copy", zc)
            big[:] = -3.0
            out = big[:cam.height]
            vrt_ctx.render(ra.Params(tile_index=1, tile_count=3, tile_rows=8, **kw), rgba=out)
            mine = np.array([(h // 8) % 3 == 1 for h in range(cam.height)])
            assert np.array_equal(out[mine], ref[mine], equal_nan=True) and (out[~mine] == -3.0).all()
            big[:] = -3.0
            vrt_ctx.render(ra.Params(tile_index=1, tile_count=3, tile_rows=8, compact_rows=True, **kw), rgba=out)
            n = int(mine.sum())
            assert np.array_equal(out[:n], ref[mine], equal_nan=True) and (out[n:] == -3.0).all()
    finally:
        vrt_ctx.set_option("zero_copy", 1)
        vrt_ctx.host_free(big)


def test_frames_into_pageable_memory_go_through_staging(vrt_ctx, oracle_mod):
    """vrt_render into an array the device cannot reach renders bands into page-locked staging and
    copies each out while the next is traced (frames of 2^18 pixels and more).  Same frame, hit
    ids, ray counts and recorded segments as the runtime's own pageable copy (option staged off),
    for a height that is no multiple of the band height; parameter errors are reported alike."""
    scene, cam, _ = _spheres(648, 483)
    kw = dict(spp=1, seed=3, max_depth=4, rng=200.0)
    vrt_ctx.set_scene(scene)
    vrt_ctx.set_camera(cam)
    px = np.array([5, 70_000, 200_123, 648 * 483 - 1], np.int32)
    out = {}
    for staged in (1, 0):
        vrt_ctx.set_option("staged", staged)
        try:
            prm = ra.Params(**kw)
            segs, counts = prm.record(px, cap=32)
            rgba = np.full((cam.height, cam.width, 4), -5.0, np.float32)
            hit = np.full((cam.height, cam.width), -5, np.int32)
            _, _, st = vrt_ctx.render(prm, rgba=rgba, hit=hit)
            out[staged] = (rgba, hit, list(st.rays), st.pixels, np.array(counts), bytes(segs))
            only = np.full_like(rgba, -5.0)
            vrt_ctx.render(ra.Params(**kw), rgba=only, hit=None, want_hit=False)
            assert np.array_equal(only, rgba, equal_nan=True)
            with pytest.raises(host.VrtError) as e:
                vrt_ctx.render(ra.Params(**dict(kw, spp=0)), rgba=rgba, hit=hit)
            assert e.value.code == abi.VRT_ERR_INVALID
        finally:
            vrt_ctx.set_option("staged", 1)
    a, b = out[1], out[0]
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1])
    assert a[2] == b[2] and a[3] == b[3] == cam.width * cam.height
    assert np.array_equal(a[4], b[4]) and a[5] == b[5]
    o_rgba, o_hit, o_rays = oracle_mod.render(scene, cam, ra.Params(**kw).c, n_threads=8)
    assert np.array_equal(a[1], o_hit) and np.array_equal(np.array(a[2], np.uint64), o_rays)


@pytest.mark.parametrize("which", ["teaching", "spheres"])
def test_frames_as_color32(vrt_ctx, oracle_mod, which):
    """VRT_OUT_RGBA8: the frame as the RGBA32 texture of the player holds it after SetPixels + Apply
    (Assets/Scripts/RayTracer/RayTracerUnity.cs:472-475), 4 bytes per pixel.  Every way a frame
    reaches the caller (pageable memory, page-locked memory written by the kernel, staged bands,
    strip sets, non-blocking, device memory, both kernel organisations, several GPUs) must deliver
    the quantisation of its own float frame bit for bit, and stay within one step of the
    quantised oracle frame (the float frames differ by up to 1/255)."""
    import torch
    if which == "teaching":
        scene, cam, kw = ra.teaching_scene(), ra.teaching_camera(640, 411), dict(spp=1, max_depth=5)   # 2^18 pixels: staged bands
    else:
        scene, cam, kw = _spheres()
    vrt_ctx.set_scene(scene)
    vrt_ctx.set_camera(cam)
    H, W = cam.height, cam.width
    f32, hit, st = vrt_ctx.render(ra.Params(**kw))
    want = oracle_mod.rgba8(f32)
    kw8 = dict(kw, flags=abi.VRT_OUT_RGBA8)
    got, got_hit, st8 = vrt_ctx.render(ra.Params(**kw8))
    assert got.dtype == np.uint8 and np.array_equal(got, want) and np.array_equal(got_hit, hit)
    assert list(st8.rays) == list(st.rays)
    if which == "spheres":
        o = oracle_mod.rgba8(_oracle_frame(oracle_mod, "rgba8", scene, cam, kw)[0])
        assert np.abs(got.astype(np.int32) - o.astype(np.int32)).max() <= 1
    # page-locked buffer inside a larger allocation: written by the kernel, nothing beside it touched
    big = vrt_ctx.host_alloc((H + 4, W, 4), np.uint8)
    try:
        big[:] = 7
        out = big[2:2 + H]
        vrt_ctx.render(ra.Params(**kw8), rgba=out, want_hit=False)
        assert np.array_equal(out, want) and (big[:2] == 7).all() and (big[2 + H:] == 7).all()
        # strip sets, non-blocking
        big[:] = 7
        vrt_ctx.render_begin_parts(ra.Params(tile_rows=8, **kw8), out, None, 3)
        vrt_ctx.render_end()
        assert np.array_equal(out, want)
    finally:
        vrt_ctx.host_free(big)
    page = np.full((H, W, 4), 7, np.uint8)
    vrt_ctx.render_begin(ra.Params(tile_index=1, tile_count=2, tile_rows=8, **kw8), page)
    vrt_ctx.render_end()
    mine = np.array([(h // 8) % 2 == 1 for h in range(H)])
    assert np.array_equal(page[mine], want[mine]) and (page[~mine] == 7).all()
    # wavefront organisation, device memory
    wf, _, _ = vrt_ctx.render(ra.Params(mode=abi.VRT_MODE_WAVEFRONT, **kw8))
    assert np.array_equal(wf, want)
    dev = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    vrt_ctx.render_device(ra.Params(**kw8), dev.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy().view(np.uint8).reshape(H, W, 4), want)
    # several GPUs, delivered per GPU and gathered
    m = host.Context(devices=[0, 0])
    try:
        m.set_scene(scene)
        m.set_camera(cam)
        for gather in (0, 1):
            m.set_option("multi_gather", gather)
            g8, g_hit, _ = m.render(ra.Params(**kw8))
            assert np.array_equal(g8, want) and np.array_equal(g_hit, hit)
    finally:
        m.close()
    with pytest.raises(host.VrtError) as e:
        vrt_ctx.render(ra.Params(**dict(kw, flags=abi.VRT_OUT_RGBA8 | abi.VRT_OUT_LINEAR_RGB)))
    assert e.value.code == abi.VRT_ERR_INVALID
