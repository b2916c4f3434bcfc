"""MatteMaterial and TransparentMaterial (defined by the reference, instantiated by none of its
scenes) on the GPU against the oracle: whole frames in both kernel organisations, per-pixel ray
trees, and the total-internal-reflection rule that looks at the BTDF's constructor direction."""
import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import abi, host

pytestmark = pytest.mark.gpu
TOL = 1.0 / 255.0


def _compare(ctx, oracle_mod, scene, cam, prm):
    ctx.set_scene(scene)
    ctx.set_camera(cam)
    rgba, hit, st = ctx.render(prm)
    o_rgba, o_hit, o_rays = oracle_mod.render(scene, cam, prm.c, n_threads=8)
    assert np.array_equal(hit, o_hit)
    assert np.array_equal(np.array(st.rays[:], np.uint64), o_rays)
    assert np.array_equal(np.isnan(rgba), np.isnan(o_rgba))
    assert np.nanmax(np.abs(rgba - o_rgba)) <= TOL
    return rgba, st


@pytest.mark.parametrize("mode", [abi.VRT_MODE_MEGAKERNEL, abi.VRT_MODE_WAVEFRONT])
@pytest.mark.parametrize("ray_dir", [(0.0, 0.0, 0.0), (0.0, 0.99, 0.14), (1.0, 0.2, 0.0)])
def test_matte_and_transparent_frames_match_oracle(vrt_ctx, oracle_mod, mode, ray_dir):
    scene, cam = ra.teaching_scene_all_materials(ray_dir), ra.teaching_camera(160, 90)
    offs = host.sampler_table(abi.VRT_SAMPLER_JITTERED, 4, 5, 0.01, 0.01, seed=4)
    rgba, st = _compare(vrt_ctx, oracle_mod, scene, cam, ra.Params(spp=4, offsets=offs, seed=2, max_depth=6, mode=mode))
    assert st.rays[abi.VRT_RAY_REFLECT] > 0
    if ray_dir == (0.0, 0.0, 0.0):
        assert st.rays[abi.VRT_RAY_TRANSMIT] > 0


def test_matte_and_transparent_with_fix_flags_and_tree(vrt_ctx, oracle_mod):
    """The same materials on a scene large enough for the tree, with every quirk switch on."""
    scene = ra.random_spheres_scene(n=600, seed=12)
    t = scene.materials["type"]
    t[t == abi.VRT_MAT_DIELECTRIC] = abi.VRT_MAT_TRANSPARENT
    t[::7] = abi.VRT_MAT_MATTE
    cam = ra.random_spheres_camera(120, 68)
    _compare(vrt_ctx, oracle_mod, scene, cam,
             ra.Params(spp=2, offsets=host.sampler_table(abi.VRT_SAMPLER_RANDOM, 2, 3, 0.02, 0.02, seed=9), seed=1,
                       max_depth=5, rng=200.0,
                       flags=abi.VRT_FIX_AO_BASIS | abi.VRT_FIX_DIELECTRIC | abi.VRT_FIX_PRIMARY_SKY))


def test_ray_trees_of_the_new_materials(vrt_ctx, oracle_mod):
    """Recorded segments (origin, end, kind, depth, primitive) of pixels on the matte and on the
    transparent sphere are the oracle's, in its depth-first order: reflection subtree, then transmission."""
    scene, cam = ra.teaching_scene_all_materials(), ra.teaching_camera(160, 90)
    vrt_ctx.set_scene(scene)
    vrt_ctx.set_camera(cam)
    prm = ra.Params(spp=1, max_depth=4)
    _, hit, _ = vrt_ctx.render(prm)
    kinds = set()
    for prim in (1, 2, 6):   # transparent sphere, matte blue, matte green
        hs, vs = np.nonzero(hit == prim)
        assert len(hs) > 0
        for k in (0, len(hs) // 2, len(hs) - 1):
            h, v = int(hs[k]), int(vs[k])
            rgba, hid, segs, n = vrt_ctx.trace_pixel(prm, h, v, -1, 128)
            o_rgba, o_hid, o_segs, o_n, _ = oracle_mod.trace_pixel(scene, cam, prm.c, h, v, -1, 128)
            assert hid == o_hid == prim and n == o_n and n <= 128
            assert bytes(segs)[:n * 40] == bytes(o_segs)[:n * 40]
            assert np.nanmax(np.abs(rgba - o_rgba)) <= TOL
            kinds |= {segs[i].kind for i in range(n)}
    assert abi.VRT_RAY_TRANSMIT in kinds and abi.VRT_RAY_REFLECT in kinds


def test_bad_material_class_is_rejected(vrt_ctx):
    scene = ra.teaching_scene()
    scene.materials["type"][0] = 6
    with pytest.raises(host.VrtError) as e:
        vrt_ctx.set_scene(scene)
    assert e.value.code == abi.VRT_ERR_INVALID
    vrt_ctx.set_scene(ra.teaching_scene())
