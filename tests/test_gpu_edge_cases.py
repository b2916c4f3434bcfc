"""Edge cases on the GPU against the oracle: smallest/degenerate scenes, capacity limits,
coincident primitives (ties by id, duplicate Morton codes in the LBVH), deep recursion."""
import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import abi, host

pytestmark = pytest.mark.gpu
TOL = 1.0 / 255.0


def _cmp(ctx, oracle_mod, scene, cam, prm):
    ctx.set_scene(scene)
    ctx.set_camera(cam)
    rgba, hit, st = ctx.render(prm)
    o_rgba, o_hit, o_rays = oracle_mod.render(scene, cam, prm.c, n_threads=8)
    assert np.array_equal(hit, o_hit)
    assert np.array_equal(np.array(st.rays[:], np.uint64), o_rays)
    both_nan = np.isnan(rgba) & np.isnan(o_rgba)
    err = np.abs(np.where(both_nan, 0, rgba) - np.where(both_nan, 0, o_rgba))
    assert not np.isnan(err).any() and err.max() <= TOL
    return rgba, hit, st


def _cam(w=64, h=48):
    return ra.make_camera((0, 1, -6), (-2, -0.5, -3), (4, 0, 0), (0, 3, 0), w, h)


def test_single_primitive_no_lights(vrt_ctx, oracle_mod):
    s = ra.Scene()
    m = s.add_material("Metal", abi.VRT_MAT_METAL, (0.9, 0.9, 0.9))
    s.add_sphere((0, 1, 0), 1.0, m)
    rgba, hit, st = _cmp(vrt_ctx, oracle_mod, s, _cam(), ra.Params(spp=1, max_depth=15))
    assert st.rays[abi.VRT_RAY_SHADOW] == 0 and st.bvh_nodes == 0
    assert (hit == 0).any() and (hit == -1).any()


def test_max_lights_and_too_many(vrt_ctx, oracle_mod):
    s = ra.Scene()
    m = s.add_material("Red", abi.VRT_MAT_SOLID, (1, 0.2, 0.2))
    g = s.add_material("Dielectric", abi.VRT_MAT_DIELECTRIC, (1, 1, 1))
    s.add_sphere((0, 1, 0), 1.0, g)
    s.add_box((0, -0.5, 0), (5, 0.5, 5), m)
    for k in range(8):
        a = 2 * np.pi * k / 8
        s.add_light(abi.VRT_LIGHT_POINT, (4 * np.cos(a), 5, 4 * np.sin(a)), ls=0.3, cast_shadows=bool(k % 2))
    _cmp(vrt_ctx, oracle_mod, s, _cam(), ra.Params(spp=1, max_depth=4))
    s.add_light(abi.VRT_LIGHT_POINT, (0, 9, 0))
    with pytest.raises(host.VrtError) as e:
        vrt_ctx.set_scene(s)
    assert e.value.code == abi.VRT_ERR_CAPACITY


def test_bvh_threshold_and_coincident_primitives(vrt_ctx, oracle_mod):
    """49 primitives is the smallest scene that gets an LBVH; identical spheres share a centroid
    (duplicate Morton codes) and tie on t, which must go to the lowest id."""
    s = ra.Scene()
    m = s.add_material("Metal", abi.VRT_MAT_METAL, (0.8, 0.8, 0.9))
    b = s.add_material("Blue", abi.VRT_MAT_SOLID, (0.1, 0.3, 1.0))
    rng = np.random.default_rng(1)
    for k in range(12):
        c = (rng.uniform(-2, 2), rng.uniform(0.3, 2), rng.uniform(-1, 3))
        for _ in range(4):  # four identical copies
            s.add_sphere(c, 0.4, m if k % 2 else b)
    s.add_box((0, -0.5, 0), (6, 0.5, 6), b)
    s.add_light(abi.VRT_LIGHT_POINT, (0, 8, -2), cast_shadows=True)
    assert s.n_prims == 49
    rgba, hit, st = _cmp(vrt_ctx, oracle_mod, s, _cam(96, 72), ra.Params(spp=1, max_depth=6))
    assert st.bvh_nodes == 48
    assert set(np.unique(hit[hit >= 0]) % 4) <= {0}  # only the first copy of each sphere (and the box, id 48)


def test_degenerate_geometry(vrt_ctx, oracle_mod):
    s = ra.Scene()
    m = s.add_material("Green", abi.VRT_MAT_SOLID, (0.2, 1.0, 0.3))
    s.add_sphere((0, 1, 0), 0.0, m)                      # zero radius
    s.add_box((1, 1, 0), (0.5, 0.0, 0.5), m)             # flat box
    verts = [(-1, 0, 2), (0, 2, 2), (1, 0, 2), (0, 0, 0), (0, 0, 0), (0, 0, 0), (-3, 0, 1), (-3, 0, 1), (-2, 1, 1)]
    s.set_mesh(verts, [(0, 1, 2), (3, 4, 5), (6, 7, 8)])  # one real triangle, two zero-area ones
    s.add_light(abi.VRT_LIGHT_DIRECTIONAL, (0, 0.6, -0.8), cast_shadows=True)
    _cmp(vrt_ctx, oracle_mod, s, _cam(), ra.Params(spp=1, max_depth=3))


def test_large_spp_and_depth(vrt_ctx, oracle_mod):
    offs = host.sampler_table(abi.VRT_SAMPLER_RANDOM, 256, 3, 0.05, 0.05, seed=9)
    _cmp(vrt_ctx, oracle_mod, ra.teaching_scene(), ra.teaching_camera(24, 16),
         ra.Params(spp=256, offsets=offs, seed=2, max_depth=15))


def test_camera_inside_a_collider_and_zero_viewport(vrt_ctx, oracle_mod):
    s = ra.teaching_scene()
    # origin inside the backdrop box: that collider is never reported
    cam = ra.make_camera((7.417, 2.0, 2.2), (6.0, 1.5, 3.0), (0, 0, -1.5), (0, 1, 0), 40, 30)
    _cmp(vrt_ctx, oracle_mod, s, cam, ra.Params(spp=1, max_depth=3))
    # degenerate viewport: every direction is zero, nothing is cast
    cam0 = ra.make_camera((1, 2, 3), (1, 2, 3), (0, 0, 0), (0, 0, 0), 8, 8)
    rgba, hit, st = _cmp(vrt_ctx, oracle_mod, s, cam0, ra.Params(spp=1, max_depth=3))
    assert sum(st.rays) == 0 and np.all(hit == -1)


def test_geometric_progression_of_primitives(vrt_ctx, oracle_mod):
    """Spheres whose centres and radii shrink geometrically towards a point: every binned split of
    the surface-area build peels a few of them off, so the tree wants to be hundreds of levels
    deep; the builder bounds its depth (median splits beyond 48 levels) and the frame must still be
    the oracle's, whatever the builder."""
    s = ra.Scene()
    m = s.add_material("Metal", abi.VRT_MAT_METAL, (0.9, 0.8, 0.7))
    b = s.add_material("Blue", abi.VRT_MAT_SOLID, (0.1, 0.3, 1.0))
    for k in range(400):
        f = 0.9 ** k
        s.add_sphere((-1.8 + 3.6 * f, 0.2 + 1.5 * f, 0.5), 0.35 * f, m if k % 3 else b)
    s.add_box((0, -0.5, 0), (6, 0.5, 6), b)
    s.add_light(abi.VRT_LIGHT_POINT, (0, 8, -2), cast_shadows=True)
    prm = ra.Params(spp=1, max_depth=4)
    ref, ref_hit, st = _cmp(vrt_ctx, oracle_mod, s, _cam(96, 72), prm)
    vrt_ctx.set_option("builder", 0)
    try:
        vrt_ctx.set_scene(s)
        got, got_hit, _ = vrt_ctx.render(prm)
    finally:
        vrt_ctx.set_option("builder", -1)
    assert np.array_equal(got, ref, equal_nan=True) and np.array_equal(got_hit, ref_hit)


def test_large_nodes_of_the_device_build(vrt_ctx, oracle_mod):
    """Nodes of more than 4 096 primitives are split by many blocks (bins, counts, stable scatter
    in separate launches).  Two scenes that push that path into its corners: 5 000 coincident
    spheres (no usable split: large nodes are halved by index, level after level) next to a
    spread-out set, and one far outlier beside 6 000 clustered spheres (a large node whose child
    is a single primitive).  The frames must be the oracle's and those of the host-built tree."""
    rng = np.random.default_rng(11)
    for outlier in (False, True):
        s = ra.Scene()
        red = s.add_material("Red", abi.VRT_MAT_SOLID, (1.0, 0.2, 0.1))
        met = s.add_material("Metal", abi.VRT_MAT_METAL, (0.9, 0.9, 0.9))
        if outlier:
            s.add_sphere((25.0, 0.5, 20.0), 0.5, met)
            for p in rng.uniform((-1.5, 0.0, -0.5), (1.5, 2.0, 1.5), (6000, 3)):
                s.add_sphere(tuple(p), 0.03, red)
        else:
            for _ in range(5000):
                s.add_sphere((0.5, 0.8, 0.5), 0.4, met)
            for p in rng.uniform((-2.0, 0.0, -0.5), (2.0, 2.0, 2.0), (1500, 3)):
                s.add_sphere(tuple(p), 0.05, red)
        s.add_box((0, -0.5, 0), (8, 0.5, 8), red)
        s.add_light(abi.VRT_LIGHT_POINT, (0, 8, -2), cast_shadows=True)
        prm = ra.Params(spp=1, max_depth=3)
        ref, ref_hit, st = _cmp(vrt_ctx, oracle_mod, s, _cam(48, 36), prm)
        assert st.bvh_nodes > 0
        vrt_ctx.set_option("builder", 2)   # the same build on the host
        try:
            vrt_ctx.set_scene(s)
            got, got_hit, _ = vrt_ctx.render(prm)
        finally:
            vrt_ctx.set_option("builder", -1)
        assert np.array_equal(got, ref, equal_nan=True) and np.array_equal(got_hit, ref_hit)


def _amid_scene(n_spheres, seed):
    """A small tree scene (49-90 primitives) whose occlusion rays leave a wave in every direction:
    a box whose top and +x faces are seen together (ambient-occlusion rays along (0,1,0) and
    (1,0,0): two infinite reciprocals each), and point lights that sit amid the spheres, just above
    the ground, so that the shadow rays of one 8x8-pixel wave differ in sign on all three axes."""
    s = ra.Scene()
    met = s.add_material("Metal", abi.VRT_MAT_METAL, (0.9, 0.85, 0.8))
    red = s.add_material("Red", abi.VRT_MAT_SOLID, (1.0, 0.25, 0.2))
    gls = s.add_material("Dielectric", abi.VRT_MAT_DIELECTRIC, (1, 1, 1))
    rng = np.random.default_rng(seed)
    for k in range(n_spheres):
        c = (rng.uniform(-2.2, 2.2), rng.uniform(0.25, 1.6), rng.uniform(-1.0, 3.0))
        s.add_sphere(c, rng.uniform(0.12, 0.3), (met, red, gls)[k % 3])
    s.add_box((-1.2, 0.3, -0.8), (0.5, 0.3, 0.5), red)      # top and +x faces towards the camera
    s.add_box((1.4, 0.2, 0.2), (0.4, 0.2, 0.6), met)
    s.add_box((0, -0.5, 1.0), (7, 0.5, 7), red)             # ground: kept out of the tree
    s.add_light(abi.VRT_LIGHT_POINT, (0.1, 0.12, 0.6), ls=0.7, cast_shadows=True)   # amid the spheres
    s.add_light(abi.VRT_LIGHT_POINT, (-0.9, 0.9, 0.1), ls=0.5, cast_shadows=True)   # beside the box
    s.add_light(abi.VRT_LIGHT_DIRECTIONAL, (0.3, 0.8, -0.5), ls=0.4, cast_shadows=False)
    return s


@pytest.mark.parametrize("n_spheres", [46, 87])
def test_small_tree_with_rays_in_every_direction(vrt_ctx, vrt_opts, oracle_mod, n_spheres):
    """Waves whose occlusion rays constrain no axis of the shared candidate collection (directions of
    both signs, or parallel to an axis, on x, y and z at once): then only its code tells an unused
    child slot of a 4-wide node from a leaf.  49 and 90 primitives: the whole tree fits the
    candidate list, so nothing falls back to the per-lane traversal."""
    s = _amid_scene(n_spheres, seed=5)
    assert s.n_prims == n_spheres + 3
    cam = ra.make_camera((0.3, 2.6, -5.5), (-2.2, 0.4, -2.5), (4.4, 0, 0), (0, 3.0, -0.9), 96, 72)
    offs = host.sampler_table(abi.VRT_SAMPLER_JITTERED, 4, 5, 0.04, 0.04, seed=3)
    for flags in (0, abi.VRT_FIX_AO_BASIS, abi.VRT_FIX_AO_BASIS | abi.VRT_FIX_DIELECTRIC):
        for slab_fma in (1, 0):
            vrt_opts("slab_fma", slab_fma)
            rgba, hit, st = _cmp(vrt_ctx, oracle_mod, s, cam, ra.Params(spp=1, max_depth=4, flags=flags))
            assert st.bvh_nodes == s.n_prims - 1 and st.rays[abi.VRT_RAY_SHADOW] > 0 and st.rays[abi.VRT_RAY_AO] > 0
            _cmp(vrt_ctx, oracle_mod, s, cam, ra.Params(spp=4, offsets=offs, seed=1, max_depth=4, flags=flags))
    vrt_opts("sample_shift", 0)   # 8x8-pixel waves with four samples looped per lane
    _cmp(vrt_ctx, oracle_mod, s, cam, ra.Params(spp=4, offsets=offs, seed=1, max_depth=4, flags=abi.VRT_FIX_AO_BASIS))


@pytest.mark.parametrize("n_spheres", [0, 1, 2, 3, 4, 5, 8, 9])
@pytest.mark.parametrize("n_boxes", [0, 1, 2])
def test_brute_force_loop_at_every_group_remainder(vrt_ctx, oracle_mod, n_spheres, n_boxes):
    """The brute-force loop takes its spheres in groups and the library pads the sphere arrays with
    spheres no ray can hit: every remainder of the group size, with and without boxes behind them
    (whose camera terms sit after the padded spheres), mirrors and a shadow-casting point light so
    that occlusion queries, bounces from spheres and from boxes and the skip id are all in play."""
    if n_spheres == 0 and n_boxes == 0:
        pytest.skip("an empty scene is refused")
    rng = np.random.default_rng(100 * n_spheres + n_boxes)
    s = ra.Scene()
    metal = s.add_material("Metal", abi.VRT_MAT_METAL, (0.9, 0.9, 0.9))
    red = s.add_material("Red", abi.VRT_MAT_SOLID, (0.9, 0.2, 0.2))
    wood = s.add_material("Wood", abi.VRT_MAT_SOLID, (0.6, 0.4, 0.2))
    for k in range(n_spheres):
        c = (float(rng.uniform(-1.8, 1.8)), float(rng.uniform(0.4, 2.2)), float(rng.uniform(-1.0, 2.0)))
        s.add_sphere(c, float(rng.uniform(0.25, 0.6)), metal if k % 2 == 0 else red)
    if n_boxes >= 1:
        s.add_box((0.0, -0.1, 0.5), (3.0, 0.1, 3.0), wood)
    if n_boxes >= 2:
        s.add_box((1.2, 0.5, 1.5), (0.4, 0.5, 0.4), metal)
    s.add_light(abi.VRT_LIGHT_POINT, (1.0, 4.0, -2.0), ls=1.0, cast_shadows=True)
    s.add_light(abi.VRT_LIGHT_DIRECTIONAL, (0.3, -1.0, 0.2), ls=0.5, cast_shadows=False)
    rgba, hit, st = _cmp(vrt_ctx, oracle_mod, s, _cam(96, 72), ra.Params(spp=1, max_depth=5))
    assert st.bvh_nodes == 0
    assert hit.max() < n_spheres + n_boxes

