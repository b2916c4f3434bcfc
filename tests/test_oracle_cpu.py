"""CPU-only checks: the oracle against the known answers the reference holds and against
hand-derived values; host logic (sampler tables, pixel->set hash); the C-ABI library loads and
exports every declared symbol.  No GPU needed."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import raytracing_amd as ra
from raytracing_amd import abi, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sky_colour_known_answer_from_reference_tests(oracle_mod):
    # Assets/Scripts/Tests/RayTracerTests.cs:48-54: CreateNonHitColor(Vector3.zero) has
    # r > 0.5, g > 0.7, |b - 1| < 1e-5; the formula gives exactly (0.75, 0.85, 1.0).
    c = oracle_mod.sky_color((0, 0, 0))
    assert c[0] > 0.5 and c[1] > 0.7 and abs(c[2] - 1.0) < 1e-5
    assert np.allclose(c, (0.75, 0.85, 1.0, 1.0), atol=1e-7)
    up = oracle_mod.sky_color((0, 5, 0))
    assert np.allclose(up[:3], (0.5, 0.7, 1.0), atol=1e-7)


def test_tone_map_max_to_one(oracle_mod):
    assert np.array_equal(oracle_mod.process_color((0.2, 0.4, 0.9)), np.float32([0.2, 0.4, 0.9, 1.0]))
    c = oracle_mod.process_color((2.0, 1.0, 0.5))
    assert np.allclose(c, (1.0, 0.5, 0.25, 0.5))  # Color/float divides alpha too


def test_camera_direction_shipped_viewport(oracle_mod):
    cam = ra.teaching_camera(200, 200)
    d = oracle_mod.camera_dir(cam, 0, 0)
    assert np.allclose(d, (1.55, -0.45, 1.0), atol=1e-6)       # P0 - O
    d = oracle_mod.camera_dir(cam, 100, 50)                    # h -> Y (vertical), v -> X
    assert np.allclose(d, (1.55, -0.45 + 0.5, 1.0 - 2.0 * 0.25), atol=1e-6)


def test_hammersley_shipped_offsets(oracle_mod):
    # SampleSize 50, vStep 0.005: Phi's argument j*vStep stays below 0.5, so y == 0 and
    # x = j * 0.01 / 50 (HammersleySampler.cs:17,22-35)
    t = oracle_mod.sampler_table(abi.VRT_SAMPLER_HAMMERSLEY, 50, 83, 0.01, 0.005, 1)
    assert np.all(t[..., 1] == 0)
    expect = np.sort((np.arange(50, dtype=np.float32) * np.float32(0.01)) / np.float32(50))
    for s in range(83):
        assert np.array_equal(np.sort(t[s, :, 0]), expect)


def test_multijittered_is_all_zero(oracle_mod):
    assert not oracle_mod.sampler_table(abi.VRT_SAMPLER_MULTIJITTERED, 16, 3, 0.01, 0.01, 5).any()


@pytest.mark.parametrize("sampler", range(6))
@pytest.mark.parametrize("n,sets", [(1, 1), (9, 4), (50, 83), (10, 3)])
def test_sampler_tables_product_equals_oracle(oracle_mod, sampler, n, sets):
    a = host.sampler_table(sampler, n, sets, 0.01, 0.005, seed=1234 + sampler)
    b = oracle_mod.sampler_table(sampler, n, sets, 0.01, 0.005, 1234 + sampler)
    assert np.array_equal(a, b)
    assert np.all(a >= 0) and np.all(a[..., 0] <= 0.01) and np.all(a[..., 1] <= 0.005)


def test_sampler_table_rejects_bad_arguments():
    lib = host.load_library()
    out = np.zeros(8, np.float32)
    assert lib.vrt_sampler_table(9, 2, 2, 0.1, 0.1, 0, out.ctypes.data) == abi.VRT_ERR_INVALID
    assert lib.vrt_sampler_table(0, 0, 2, 0.1, 0.1, 0, out.ctypes.data) == abi.VRT_ERR_INVALID
    assert lib.vrt_sampler_table(0, 2, 2, 0.1, 0.1, 0, None) == abi.VRT_ERR_INVALID


def test_pixel_set_product_equals_oracle(oracle_mod):
    for seed in (0, 7, 0xdeadbeef):
        for pix in (0, 1, 63, 40000, 2073599, 33177599):
            for sets in (1, 2, 83, 1000):
                a = host.pixel_set(seed, pix, sets)
                assert a == oracle_mod.pixel_set(seed, pix, sets)
                assert 0 <= a < sets


def test_library_exports_every_declared_symbol():
    lib = host.load_library()
    hdr = open(os.path.join(ROOT, "include", "vrt.h")).read()
    declared = set(re.findall(r"\b(vrt_[a-z_]+)\s*\(", hdr))
    assert declared == set(abi.EXPORTS), declared ^ set(abi.EXPORTS)
    for name in abi.EXPORTS:
        assert getattr(lib, name) is not None
    assert lib.vrt_abi_version() == abi.VRT_ABI_VERSION


def test_struct_sizes_match_header_layout():
    assert C.sizeof(abi.vrt_sphere) == 20 and C.sizeof(abi.vrt_box) == 64
    assert C.sizeof(abi.vrt_material) == 20 and C.sizeof(abi.vrt_path_seg) == 40
    assert C.sizeof(abi.vrt_camera) == 56
    d = abi.vrt_shading()
    host.load_library().vrt_shading_defaults(C.byref(d))
    py = ra.scenes.default_shading()
    assert bytes(d) == bytes(py)
    p = abi.vrt_params()
    host.load_library().vrt_params_defaults(C.byref(p))
    assert (p.spp, p.max_depth, p.range, p.tile_count) == (1, 10, 30.0, 1)


def test_sphere_box_triangle_hits(oracle_mod):
    s = ra.Scene()
    m = s.add_material("Red", abi.VRT_MAT_SOLID, (1, 0, 0))
    s.add_sphere((0, 0, 5), 1.0, m)
    s.add_box((0, 0, 10), (1, 1, 1), m)
    hit, t, prim, pt, n = oracle_mod.intersect(s, (0, 0, 0), (0, 0, 1))
    assert hit and prim == 0 and t == 4.0 and np.array_equal(n, (0, 0, -1))
    hit, t, prim, pt, n = oracle_mod.intersect(s, (0, 0, 0), (0, 0, 1), skip=0)
    assert hit and prim == 1 and t == 9.0 and np.array_equal(n, (0, 0, -1))
    # origin inside a collider: not reported
    hit, t, prim, _, _ = oracle_mod.intersect(s, (0, 0, 5), (0, 0, 1))
    assert hit and prim == 1
    hit, *_ = oracle_mod.intersect(s, (0, 0, 10), (0, 1, 0))
    assert not hit
    # beyond maxDistance
    hit, *_ = oracle_mod.intersect(s, (0, 0, 0), (0, 0, 1), maxd=3.9)
    assert not hit
    # zero direction: Physics.Raycast returns false
    hit, *_ = oracle_mod.intersect(s, (0, 0, 0), (0, 0, 0))
    assert not hit
    # single-sided triangle, clockwise seen from -z
    s2 = ra.Scene()
    s2.add_material("T", abi.VRT_MAT_SOLID, (1, 1, 1))
    s2.set_mesh([(-1, -1, 3), (0, 1, 3), (1, -1, 3)], [(0, 1, 2)])
    hit, t, prim, _, n = oracle_mod.intersect(s2, (0, 0, 0), (0, 0, 1))
    assert hit and t == 3.0 and np.array_equal(n, (0, 0, -1))
    hit, *_ = oracle_mod.intersect(s2, (0, 0, 6), (0, 0, -1))
    assert not hit


def test_oracle_bvh_equals_brute_force(oracle_mod):
    scene = ra.random_spheres_scene(n=300, seed=2)
    rng = np.random.default_rng(0)
    for _ in range(400):
        o = rng.uniform(-30, 30, 3) * (1, 0.2, 1) + (0, 6, 0)
        d = rng.normal(size=3)
        a = oracle_mod.intersect(scene, o, d, 200.0)
        b = oracle_mod.intersect(scene, o, d, 200.0, force_brute=True)
        assert a[0] == b[0] and a[2] == b[2] and (not a[0] or a[1] == b[1])


def test_ao_direction_rule(oracle_mod):
    """The ambient occluder's direction degenerates (no ray is cast) exactly when n.x < 0 or
    n.z < 0 or (n.y < 0 and n.y != -1); otherwise it is the normal.  The GPU code uses this
    closed form; here it is checked against the literal Pow arithmetic."""
    rng = np.random.default_rng(1)
    normals = [(0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0), (0, 0, 1), (0, 0, -1), (-0.0, 1, 0),
               (0, 1, -0.0), (1e-30, 1, 1e-30), (0.6, 0, 0.8), (0.6, -0.0, 0.8)]
    for _ in range(3000):
        v = rng.normal(size=3)
        if rng.random() < 0.3:
            v[rng.integers(3)] = 0.0
        normals.append(v / np.linalg.norm(v))
    for n in normals:
        n = np.float32(n)
        ok, d = oracle_mod.ao_direction(n)
        rule = not (n[0] < 0 or n[2] < 0 or (n[1] < 0 and n[1] != -1))
        assert ok == rule, n
        if ok:
            m = np.float32(math.sqrt(float(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])))
            assert np.allclose(d, n / m, atol=1e-6)


def test_hand_derived_shading_values(oracle_mod):
    """One solid sphere under one directional light shining straight 'up' (wi = +y), camera above:
    the top of the sphere gets ka*cd*1 + kd*cd/pi * ls (Lambert F = 0.5*cd*0.31830987)."""
    s = ra.Scene()
    m = s.add_material("Red", abi.VRT_MAT_SOLID, (1.0, 0.5, 0.25))
    s.add_sphere((0, 0, 0), 1.0, m)
    s.add_light(abi.VRT_LIGHT_DIRECTIONAL, (0, 1, 0), ls=1.0, cast_shadows=False)
    cam = ra.make_camera((0, 5, 0), (-0.5, 4, -0.5), (1, 0, 0), (0, 0, 1), 2, 2)
    prm = ra.Params(spp=1, max_depth=3)
    rgba, hit, segs, n, rays = oracle_mod.trace_pixel(s, cam, prm.c, 1, 1, -1, 16)
    assert hit == 0
    cd = np.float32([1.0, 0.5, 0.25])
    expect = np.float32(0.5) * cd + (np.float32(0.5) * cd * np.float32(0.3183098861837906715))
    assert np.allclose(rgba[:3], expect, atol=1e-6)
    # metal: mirror weight kr*cr/|n.wi| * (n.wi) = 0.75
    s.materials["type"][0] = abi.VRT_MAT_METAL
    rgba2, *_ = oracle_mod.trace_pixel(s, cam, prm.c, 1, 1, -1, 16)
    sky_up = np.float32([0.5, 0.7, 1.0])
    ph = np.float32(0.25) * cd + np.float32(0.5) * cd * np.float32(0.3183098861837906715) + np.float32(0.15)
    raw = ph + np.float32(0.75) * sky_up
    assert np.allclose(rgba2[:3], raw / max(1.0, raw.max()), atol=1e-5)
    # dielectric as shipped: always the total-internal-reflection branch, weight 1
    s.materials["type"][0] = abi.VRT_MAT_DIELECTRIC
    rgba3, *_ = oracle_mod.trace_pixel(s, cam, prm.c, 1, 1, -1, 16)
    ph = np.float32(0.5) * cd + np.float32(0.5) * cd * np.float32(0.3183098861837906715) + np.float32(0.2)
    raw = ph + sky_up
    assert np.allclose(rgba3[:3], raw / max(1.0, raw.max()), atol=1e-5)


def test_hand_derived_matte_and_transparent(oracle_mod):
    """The two material classes the reference defines but never instantiates, on the same set-up:
    MatteMaterial = ambient + Lambert without a shadow test (Material/MatteMaterial.cs:44-75);
    TransparentMaterial straight down the normal: reflection weight kr, transmission weight
    kt / ior^2 straight through (Material/TransparentMaterial.cs:40-69, BTDF/PerfectTransmitterBTDF.cs:32-49)."""
    s = ra.Scene()
    m = s.add_material("M", abi.VRT_MAT_MATTE, (1.0, 0.5, 0.25))
    s.add_sphere((0, 0, 0), 1.0, m)
    s.add_light(abi.VRT_LIGHT_DIRECTIONAL, (0, 1, 0), ls=1.0, cast_shadows=True, location=(0, 50, 0))
    cam = ra.make_camera((0, 5, 0), (-0.5, 4, -0.5), (1, 0, 0), (0, 0, 1), 2, 2)
    prm = ra.Params(spp=1, max_depth=3)
    cd = np.float32([1.0, 0.5, 0.25])
    lambert = np.float32(0.5) * cd * np.float32(0.3183098861837906715)
    rgba, hit, segs, n, rays = oracle_mod.trace_pixel(s, cam, prm.c, 1, 1, -1, 16)
    assert np.allclose(rgba[:3], np.float32(0.5) * cd + lambert, atol=1e-6)
    assert rays[abi.VRT_RAY_SHADOW] == 0 and rays[abi.VRT_RAY_AO] == 1   # no shadow ray although the light casts shadows
    # transparent, straight down the normal at the top of the sphere
    s.materials["type"][0] = abi.VRT_MAT_TRANSPARENT
    rgba3, hit3, segs3, n3, rays3 = oracle_mod.trace_pixel(s, cam, prm.c, 1, 1, -1, 16)
    sky_up, sky_down = np.float32([0.5, 0.7, 1.0]), np.float32([1.0, 1.0, 1.0])
    phong = np.float32(0.5) * cd + lambert + np.float32(0.5)          # ks * 1^2000
    raw = phong + np.float32(0.1) * sky_up + np.float32(0.9 / 2.25) * sky_down
    assert np.allclose(rgba3[:3], raw / max(1.0, raw.max()), atol=1e-5)
    assert rays3[abi.VRT_RAY_REFLECT] == 1 and rays3[abi.VRT_RAY_TRANSMIT] == 1
    # PerfectTransmitterBTDF.Tir looks at the direction it was constructed with: (0, 1, 0) makes it
    # "inside, grazing" for this normal (n.wo = -1 is not grazing) -> still refraction; (1, 0.2, 0) -> TIR
    s.shading.transp_ray_dir[:] = (1.0, 0.2, 0.0)
    rgba4, _, _, _, rays4 = oracle_mod.trace_pixel(s, cam, prm.c, 1, 1, -1, 16)
    raw = phong + sky_up                                                # reflection with weight 1
    assert np.allclose(rgba4[:3], raw / max(1.0, raw.max()), atol=1e-5)
    assert rays4[abi.VRT_RAY_REFLECT] == 1 and rays4[abi.VRT_RAY_TRANSMIT] == 0


def test_primary_miss_is_black_and_ray_classes(oracle_mod):
    scene, cam = ra.teaching_scene(), ra.teaching_camera(40, 40)
    prm = ra.Params(spp=1, max_depth=2)
    rgba, hit, rays = oracle_mod.render(scene, cam, prm.c)
    assert np.all(rgba[hit < 0][:, :3] == 0)
    assert rays[abi.VRT_RAY_PRECHECK] == 1600
    assert rays[abi.VRT_RAY_PRIMARY] == np.count_nonzero(hit >= 0)
    assert rays[abi.VRT_RAY_TRANSMIT] == 0
    prm2 = ra.Params(spp=1, max_depth=2, flags=abi.VRT_FIX_PRIMARY_SKY)
    rgba2, _, _ = oracle_mod.render(scene, cam, prm2.c)
    assert np.all(rgba2[hit < 0][:, 2] == 1.0)


def test_oracle_omp_equals_serial(oracle_mod):
    scene, cam = ra.teaching_scene(), ra.teaching_camera(64, 40)
    offs = host.sampler_table(abi.VRT_SAMPLER_JITTERED, 4, 5, 0.01, 0.005, seed=2)
    prm = ra.Params(spp=4, offsets=offs, seed=1, max_depth=5)
    a = oracle_mod.render(scene, cam, prm.c, n_threads=1)
    b = oracle_mod.render(scene, cam, prm.c, n_threads=4)
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2])


def test_csharp_binding_agrees_with_the_header():
    """The C# adapter cannot be compiled here; at least its ABI version, flag values and the entry
    points it imports are those of include/vrt.h."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cs = open(os.path.join(root, "unity", "VrtNative.cs")).read()
    hdr = open(os.path.join(root, "include", "vrt.h")).read()
    assert int(re.search(r"AbiVersion\s*=\s*(\d+)", cs).group(1)) == int(re.search(r"#define VRT_ABI_VERSION (\d+)", hdr).group(1))
    declared = set(re.findall(r"\b(vrt_[a-z_]+)\s*\(", hdr))
    imported = set(re.findall(r"extern\s+\w+\s+(vrt_[a-z_]+)\s*\(", cs))
    assert imported and imported <= declared, imported - declared
    m = re.search(r"FixPrimarySky = (\d+), FixAoBasis = (\d+), FixDielectric = (\d+)", cs)
    assert [int(x) for x in m.groups()] == [abi.VRT_FIX_PRIMARY_SKY, abi.VRT_FIX_AO_BASIS, abi.VRT_FIX_DIELECTRIC]
    assert set(abi.EXPORTS) - {"vrt_abi_version"} <= imported | {"vrt_abi_version"}, set(abi.EXPORTS) - imported
    m = re.search(r"MatMatte = (\d+), MatTransparent = (\d+)", cs)
    assert [int(x) for x in m.groups()] == [abi.VRT_MAT_MATTE, abi.VRT_MAT_TRANSPARENT]

    # every struct: same fields in the same order, compared as a flat list of 4-byte words / pointers
    words = {"int": 1, "uint": 1, "float": 1, "Vector3": 3, "Color": 4, "IntPtr": "p", "ulong": 2, "Shading": None}

    def cs_layout(name):
        body = re.search(r"struct %s\s*\{(.*?)\}" % name, cs, re.S).group(1)
        out = []
        for typ, names in re.findall(r"public\s+(?:fixed\s+)?(\w+)\s+([^;]+);", body):
            for n in names.split(","):
                n = n.strip()
                arr = re.search(r"\[(\d+)\]", n)
                k = int(arr.group(1)) if arr else 1
                if typ == "Shading":
                    out += cs_layout("Shading")
                elif words[typ] == "p":
                    out += ["p"] * k
                else:
                    out += ["w"] * (words[typ] * k)
        return out

    def c_layout(ct):
        out = []
        for fname, ftype in ct._fields_:
            if issubclass(ftype, C.Structure):
                out += c_layout(ftype)
            elif issubclass(ftype, (C._Pointer, C.c_void_p)) or ftype is C.c_char_p:
                out += ["p"]
            else:
                out += ["w"] * (C.sizeof(ftype) // 4)
        return out

    for cs_name, ct in (("Sphere", abi.vrt_sphere), ("Box", abi.vrt_box), ("Material", abi.vrt_material),
                        ("Light", abi.vrt_light), ("Shading", abi.vrt_shading), ("Texture", abi.vrt_texture),
                        ("SceneDesc", abi.vrt_scene_desc), ("Camera", abi.vrt_camera), ("PathSeg", abi.vrt_path_seg),
                        ("Params", abi.vrt_params), ("Stats", abi.vrt_stats)):
        assert cs_layout(cs_name) == c_layout(ct), cs_name
