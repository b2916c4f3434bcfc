"""Scene / camera containers and the synthetic workloads named by the benchmark configs.

A Scene owns numpy arrays laid out exactly like the structs of include/vrt.h and hands
out a vrt_scene_desc that points into them (the arrays must outlive the desc).
Citations are relative to /root/reference/VisualRayTrace/.
"""
import ctypes as C
import math

import numpy as np

from . import abi

_SPHERE_DT = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("material", "<i4")])
_BOX_DT = np.dtype([("center", "<f4", 3), ("axis", "<f4", 9), ("half", "<f4", 3),
                    ("material", "<i4")])
_MAT_DT = np.dtype([("type", "<i4"), ("color", "<f4", 4)])
_LIGHT_DT = np.dtype([("type", "<i4"), ("vec", "<f4", 3), ("ls", "<f4"), ("color", "<f4", 4),
                      ("cast_shadows", "<i4"), ("location", "<f4", 3)])
assert _SPHERE_DT.itemsize == C.sizeof(abi.vrt_sphere)
assert _BOX_DT.itemsize == C.sizeof(abi.vrt_box)
assert _MAT_DT.itemsize == C.sizeof(abi.vrt_material)
assert _LIGHT_DT.itemsize == C.sizeof(abi.vrt_light)


def default_shading():
    """Shipped constants: Assets/Scripts/RayTracer/Util/RayTraceUtility.cs:15,43-47,73-76,
    484-489; BRDF/LambertianBRDF.cs:19; BRDF/GlossySpecularBRDF.cs:17-20.  The glass
    exponent stays 1 because DielectricMaterial.SetExp(float) is empty
    (Material/DielectricMaterial.cs:27-31)."""
    s = abi.vrt_shading()
    s.solid_ka, s.solid_kd, s.solid_ks, s.solid_exp = 0.5, 0.5, 0.0, 1.0
    s.metal_ka, s.metal_kd, s.metal_ks, s.metal_exp, s.metal_kr = 0.25, 0.5, 0.15, 100.0, 0.75
    s.metal_cr[:] = (1.0, 1.0, 1.0, 1.0)
    s.glass_ka, s.glass_kd, s.glass_ks, s.glass_exp = 0.5, 0.5, 0.2, 1.0
    s.glass_eta_in, s.glass_eta_out = 1.5, 1.0
    s.ao_min_amount[:] = (0.05, 0.05, 0.05, 1.0)
    s.ao_ls = 1.0
    s.ao_color[:] = (1.0, 1.0, 1.0, 1.0)
    # Material/MatteMaterial.cs:14-15 (two default LambertianBRDFs); Material/TransparentMaterial.cs:10-25
    # has no shipped constructor call: Phong base defaults plus host-chosen ks, exp, ior, kr, kt
    s.matte_ka, s.matte_kd = 0.5, 0.5
    s.transp_ka, s.transp_kd, s.transp_ks, s.transp_exp = 0.5, 0.5, 0.5, 2000.0
    s.transp_ior, s.transp_kr, s.transp_kt = 1.5, 0.1, 0.9
    s.transp_ray_dir[:] = (0.0, 0.0, 0.0)
    return s


class Scene:
    def __init__(self):
        self.spheres = np.zeros(0, _SPHERE_DT)
        self.boxes = np.zeros(0, _BOX_DT)
        self.vertices = np.zeros((0, 3), np.float32)
        self.indices = np.zeros((0, 3), np.uint32)
        self.tri_material = None
        self.materials = np.zeros(0, _MAT_DT)
        self.lights = np.zeros(0, _LIGHT_DT)
        self.shading = default_shading()
        self._mat_names = []
        self.textures = []      # (rgba float32 [H,W,4], wrap)
        self.mat_texture = []   # per material: texture index or -1
        self.uvs = None         # per mesh vertex (u, v)

    # ---- building
    def add_texture(self, rgba, wrap=0):
        """Material.mainTexture as a float RGBA image [H, W, 4], row 0 = Unity's y = 0;
        wrap 0 = Repeat, 1 = Clamp."""
        self.textures.append((np.ascontiguousarray(rgba, np.float32), int(wrap)))
        return len(self.textures) - 1

    def add_material(self, name, mtype, color, texture=-1):
        """`name` plays the role of the Unity material name that DetermineMaterialType
        (Util/RayTraceUtility.cs:139-152) switches on; `texture` that of mainTexture
        (Tracer/WhittedTracer.cs:55-63)."""
        while len(self.mat_texture) < len(self.materials):
            self.mat_texture.append(-1)
        self.mat_texture.append(int(texture))
        m = np.zeros(1, _MAT_DT)
        m["type"] = mtype
        col = tuple(color) + (1.0,) * (4 - len(color))
        m["color"] = col
        self.materials = np.concatenate([self.materials, m])
        self._mat_names.append(name)
        return len(self.materials) - 1

    def add_sphere(self, center, radius, material):
        s = np.zeros(1, _SPHERE_DT)
        s["center"] = center
        s["radius"] = radius
        s["material"] = material
        self.spheres = np.concatenate([self.spheres, s])

    def add_box(self, center, half, material, axes=None):
        b = np.zeros(1, _BOX_DT)
        b["center"] = center
        b["half"] = half
        b["axis"] = np.eye(3, dtype=np.float32).reshape(9) if axes is None else np.asarray(
            axes, np.float32).reshape(9)
        b["material"] = material
        self.boxes = np.concatenate([self.boxes, b])

    def add_light(self, ltype, vec, ls=1.0, color=(1, 1, 1, 1), cast_shadows=True,
                  location=(0, 0, 0)):
        l = np.zeros(1, _LIGHT_DT)
        l["type"] = ltype
        l["vec"] = vec
        l["ls"] = ls
        l["color"] = color
        l["cast_shadows"] = int(cast_shadows)
        l["location"] = location
        self.lights = np.concatenate([self.lights, l])

    def set_mesh(self, vertices, indices, tri_material=None, uvs=None):
        self.uvs = None if uvs is None else np.ascontiguousarray(uvs, np.float32).reshape(-1, 2)
        self.vertices = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
        self.tri_material = None if tri_material is None else np.ascontiguousarray(
            tri_material, np.int32)

    @property
    def n_prims(self):
        return len(self.spheres) + len(self.boxes) + len(self.indices)

    # ---- ABI view
    def desc(self):
        d = abi.vrt_scene_desc()
        self.spheres = np.ascontiguousarray(self.spheres)
        self.boxes = np.ascontiguousarray(self.boxes)
        d.n_spheres = len(self.spheres)
        d.spheres = self.spheres.ctypes.data_as(C.POINTER(abi.vrt_sphere))
        d.n_boxes = len(self.boxes)
        d.boxes = self.boxes.ctypes.data_as(C.POINTER(abi.vrt_box))
        d.n_vertices = len(self.vertices)
        d.vertices = self.vertices.ctypes.data_as(C.POINTER(C.c_float))
        d.n_triangles = len(self.indices)
        d.indices = self.indices.ctypes.data_as(C.POINTER(C.c_uint32))
        if self.tri_material is not None:
            d.tri_material = self.tri_material.ctypes.data_as(C.POINTER(C.c_int32))
        d.n_materials = len(self.materials)
        d.materials = self.materials.ctypes.data_as(C.POINTER(abi.vrt_material))
        d.n_lights = len(self.lights)
        d.lights = self.lights.ctypes.data_as(C.POINTER(abi.vrt_light))
        d.shading = self.shading
        if self.textures:
            self._tex_structs = (abi.vrt_texture * len(self.textures))()
            for k, (img, wrap) in enumerate(self.textures):
                t = self._tex_structs[k]
                t.height, t.width, t.wrap = img.shape[0], img.shape[1], wrap
                t.rgba = img.ctypes.data_as(C.POINTER(C.c_float))
            d.n_textures = len(self.textures)
            d.textures = self._tex_structs
            mt = list(self.mat_texture) + [-1] * (len(self.materials) - len(self.mat_texture))
            self._mat_tex = np.ascontiguousarray(mt[:len(self.materials)], np.int32)
            d.material_texture = self._mat_tex.ctypes.data_as(C.POINTER(C.c_int32))
        if self.uvs is not None:
            assert len(self.uvs) == len(self.vertices)
            d.uvs = self.uvs.ctypes.data_as(C.POINTER(C.c_float))
        return d


def make_camera(origin, p0, x_axis, y_axis, width, height):
    c = abi.vrt_camera()
    c.origin[:] = [float(x) for x in origin]
    c.p0[:] = [float(x) for x in p0]
    c.x_axis[:] = [float(x) for x in x_axis]
    c.y_axis[:] = [float(x) for x in y_axis]
    c.width, c.height = int(width), int(height)
    return c


# --------------------------------------------------------------------------- teaching scene

def directional_from_euler(euler_deg):
    """The vector the reference stores for a directional light: (0,0,1) rotated about the
    world X, then Y, then Z axis by the light's euler angles, each only when non-zero
    (Util/RayTraceUtility.cs:500-512).  That vector is then used as the direction *to* the
    light (Light/DirectionalLight.cs:22-25)."""
    v = np.array([0.0, 0.0, 1.0])
    ex, ey, ez = [math.radians(a % 360.0) for a in euler_deg]
    if ex != 0.0:
        c, s = math.cos(ex), math.sin(ex)
        v = np.array([v[0], c * v[1] - s * v[2], s * v[1] + c * v[2]])
    if ey != 0.0:
        c, s = math.cos(ey), math.sin(ey)
        v = np.array([c * v[0] + s * v[2], v[1], -s * v[0] + c * v[2]])
    if ez != 0.0:
        c, s = math.cos(ez), math.sin(ez)
        v = np.array([c * v[0] - s * v[1], s * v[0] + c * v[1], v[2]])
    return v.astype(np.float32)


def teaching_scene(point_light=True):
    """The diorama of Assets/Scenes/VisualRayTrace.unity flattened to world space:
    7 spheres (Assets/Prefabs/RayTrace_Sphere.prefab:34,95; resting on the floor, since the
    prefab's Rigidbody has gravity, :108), the floor/border/backdrop boxes
    (VisualRayTrace.unity:35222-35302, Assets/Prefabs/RTScene_Border.prefab:81-93) and the two
    directional lights (:6104-6126, :22985-23007).  Room walls, planks and the table (mesh
    colliders further out) are left out.  `point_light` adds the one point light with
    shadows that the benchmark's "teaching scene" asks for; the shipped scene has none."""
    s = Scene()
    metal = s.add_material("Metal", abi.VRT_MAT_METAL, (0.9528, 0.9528, 0.9528))
    blue = s.add_material("Blue", abi.VRT_MAT_SOLID, (0.0, 0.6747, 1.0))
    red = s.add_material("Red", abi.VRT_MAT_SOLID, (1.0, 0.0, 0.0))
    glass = s.add_material("Dielectric", abi.VRT_MAT_DIELECTRIC, (1.0, 1.0, 1.0))
    green = s.add_material("Green", abi.VRT_MAT_SOLID, (0.4563, 1.0, 0.0))
    lwood = s.add_material("LightBrownWood", abi.VRT_MAT_SOLID, (0.8208, 0.6273, 0.2671))
    bwood = s.add_material("BrownWood", abi.VRT_MAT_SOLID, (0.3774, 0.2403, 0.2403))
    white = s.add_material("WhiteEye", abi.VRT_MAT_SOLID, (1.0, 1.0, 1.0))
    # name, authored x/z, radius; y = floor top (1.0) + radius once the spheres have settled
    for cx, cz, r, m in ((6.5, 3.75, 0.75, metal), (6.5, 1.75, 0.75, metal),
                         (5.5, 4.0, 0.25, blue), (5.5, 3.5, 0.25, metal),
                         (5.5, 3.0, 0.25, red), (5.5, 2.5, 0.25, glass),
                         (5.5, 2.0, 0.25, green)):
        s.add_sphere((cx, 1.0 + r, cz), r, m)
    # floor: unit BoxCollider, rot Y 90 deg, local scale (5.5, 1, 2.5)
    s.add_box((6.081, 0.5, 2.268), (2.75, 0.5, 1.25), lwood,
              axes=((0, 0, -1), (0, 1, 0), (1, 0, 0)))
    s.add_box((6.081, 1.099, 4.561), (1.25, 0.099, 0.182), bwood)     # Border01
    s.add_box((6.075, 1.099, -0.344), (1.245, 0.099, 0.1375), bwood)  # Border02
    s.add_box((7.417, 2.0, 2.219), (0.083, 2.0, 2.82), white)         # Backdrop
    s.add_box((6.085, 2.5, 4.934), (1.25, 1.5, 0.103), white)         # SkyLeft
    s.add_light(abi.VRT_LIGHT_DIRECTIONAL, directional_from_euler((90.0, 0.0, 0.0)),
                ls=1.0, cast_shadows=False)
    s.add_light(abi.VRT_LIGHT_DIRECTIONAL,
                directional_from_euler((53.461, 45.592, -247.916)), ls=1.0, cast_shadows=False)
    if point_light:
        s.add_light(abi.VRT_LIGHT_POINT, (4.5, 5.3, 2.5), ls=1.0, cast_shadows=True)
    return s


def teaching_scene_all_materials(transparent_ray_dir=(0.0, 0.0, 0.0)):
    """The teaching diorama with the two material classes the reference defines but never
    instantiates put on three of its spheres: Material/MatteMaterial.cs on the blue and the green
    one, Material/TransparentMaterial.cs on the large front sphere.  transparent_ray_dir is the
    direction PerfectTransmitterBTDF keeps from its constructor (BTDF/PerfectTransmitterBTDF.cs:17-20)."""
    s = teaching_scene()
    s.materials["type"][1] = abi.VRT_MAT_MATTE        # Blue
    s.materials["type"][4] = abi.VRT_MAT_MATTE        # Green
    glassy = s.add_material("Transparent", abi.VRT_MAT_TRANSPARENT, (0.65, 0.45, 0.0))
    s.spheres["material"][1] = glassy
    s.shading.transp_ray_dir[:] = [float(x) for x in transparent_ray_dir]
    return s


def teaching_camera(width=200, height=200):
    """Eye/RayCaster at (1.45, 2, 2.5) (VisualRayTrace.unity:45575-45590,13735-13748) and the
    viewport quad at (3, 2.05, 2.5), rot Y 90 deg, scale (2,1,1) (:2652-2669): P0 = (3, 1.55,
    3.5), X = (0,0,-2), Y = (0,1,0).  For a non-square frame the quad is widened so that
    pixels stay as tall as the shipped 200x200 ones are relative to the 1-unit height, i.e.
    width_units = height_units * width/height for width != height."""
    if width == height:
        wu = 2.0
    else:
        wu = 1.0 * width / height
    return make_camera((1.45, 2.0, 2.5), (3.0, 1.55, 2.5 + wu / 2), (0.0, 0.0, -wu),
                       (0.0, 1.0, 0.0), width, height)


# --------------------------------------------------------------------------- textured quad

def checker_texture(w, h, a=(1.0, 0.0, 0.0, 1.0), b=(0.0, 0.0, 1.0, 1.0)):
    """w x h checkerboard; texel (0,0) — the one untextured-coordinate colliders see — is
    (0.25, 0.5, 0.75)."""
    img = np.zeros((h, w, 4), np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    img[(xx + yy) % 2 == 0] = a
    img[(xx + yy) % 2 == 1] = b
    img[0, 0] = (0.25, 0.5, 0.75, 1.0)
    return img


def textured_quad_scene(wrap=0, uv_scale=1.0):
    """A textured quad (two triangles) facing -z behind a metal sphere, a textured sphere and a
    solid floor box, one point light: exercises Tracer/WhittedTracer.cs:55-63 directly and in
    reflections.  uv_scale > 1 pushes the texture coordinates outside [0, 1]."""
    s = Scene()
    tex = s.add_texture(checker_texture(8, 4), wrap)
    quad = s.add_material("Poster", abi.VRT_MAT_SOLID, (1, 1, 1), texture=tex)
    metal = s.add_material("Metal", abi.VRT_MAT_METAL, (0.9, 0.9, 0.9))
    floor = s.add_material("Wood", abi.VRT_MAT_SOLID, (0.8, 0.6, 0.3))
    ball = s.add_material("Ball", abi.VRT_MAT_SOLID, (1, 1, 1), texture=tex)
    v = np.array([(-3, 0, 3), (3, 0, 3), (3, 4, 3), (-3, 4, 3)], np.float32)
    uv = np.array([(0, 0), (1, 0), (1, 1), (0, 1)], np.float32) * uv_scale - (uv_scale - 1) / 2
    s.set_mesh(v, np.array([(0, 2, 1), (0, 3, 2)], np.uint32), np.array([quad, quad], np.int32), uvs=uv)
    s.add_sphere((0.0, 1.0, 0.0), 1.0, metal)
    s.add_sphere((2.2, 0.5, 0.5), 0.5, ball)
    s.add_box((0, -0.5, 0), (6, 0.5, 6), floor)
    s.add_light(abi.VRT_LIGHT_POINT, (2, 6, -4), ls=1.0, cast_shadows=True)
    return s


def textured_quad_camera(w=96, h=64):
    return make_camera((0, 2, -7), (-3, 0, -4), (6, 0, 0), (0, 4, 0), w, h)


# --------------------------------------------------------------------------- 10k spheres

def random_spheres_scene(n=10000, seed=0xC0FFEE):
    """n spheres with centres uniform in a 60 x 8 x 60 box above a ground box, radii
    U(0.1,0.5), 60 % solid / 30 % metal / 10 % dielectric, colours U(0,1)^3, one point
    light with shadows.  Generator: numpy.random.Generator(PCG64(seed)), draws in the
    order centres (n,3), radii (n), class (n), colours (n,3)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cen = rng.random((n, 3), dtype=np.float64)
    rad = rng.random(n, dtype=np.float64)
    cls = rng.random(n, dtype=np.float64)
    colr = rng.random((n, 3), dtype=np.float64)
    s = Scene()
    ground = s.add_material("Ground", abi.VRT_MAT_SOLID, (0.6, 0.6, 0.55))
    mats = np.zeros(n, _MAT_DT)
    mats["type"] = np.where(cls < 0.6, abi.VRT_MAT_SOLID,
                            np.where(cls < 0.9, abi.VRT_MAT_METAL, abi.VRT_MAT_DIELECTRIC))
    mats["color"][:, :3] = colr.astype(np.float32)
    mats["color"][:, 3] = 1.0
    s.materials = np.concatenate([s.materials, mats])
    sp = np.zeros(n, _SPHERE_DT)
    sp["center"][:, 0] = (cen[:, 0] * 60.0 - 30.0).astype(np.float32)
    sp["center"][:, 1] = (cen[:, 1] * 8.0 + 0.6).astype(np.float32)
    sp["center"][:, 2] = (cen[:, 2] * 60.0 - 30.0).astype(np.float32)
    sp["radius"] = (0.1 + 0.4 * rad).astype(np.float32)
    sp["material"] = np.arange(1, n + 1, dtype=np.int32)
    s.spheres = sp
    s.add_box((0.0, -0.5, 0.0), (40.0, 0.5, 40.0), ground)
    s.add_light(abi.VRT_LIGHT_POINT, (0.0, 40.0, -10.0), ls=1.0, cast_shadows=True)
    return s


def random_spheres_camera(width=3840, height=2160):
    """Looks along +z and slightly down over the sphere field; range must cover the scene."""
    o = np.array([0.0, 14.0, -42.0])
    fwd = np.array([0.0, 3.0, -5.0]) - o
    fwd /= np.linalg.norm(fwd)
    centre = o + 4.0 * fwd
    right = np.cross([0.0, 1.0, 0.0], fwd)
    right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    hu = 2.4
    wu = hu * width / height
    p0 = centre - right * wu / 2 - up * hu / 2
    return make_camera(o, p0, right * wu, up * hu, width, height)


# --------------------------------------------------------------------------- 1M triangles

def tessellated_mesh_scene(quads=708, seed=1):
    """A height field of quads x quads cells (2 triangles each; 708 -> 1,002,528 triangles)
    over [-20,20]^2, displaced by a fixed sum of sines whose phases come from
    numpy.random.Generator(PCG64(seed)); alternating bands of solid and metal submeshes;
    one point light with shadows.  Triangles wind clockwise seen from above (Unity)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ph = rng.random(6) * 2 * np.pi
    g = np.linspace(-20.0, 20.0, quads + 1)
    x, z = np.meshgrid(g, g, indexing="ij")
    y = (0.8 * np.sin(0.35 * x + ph[0]) * np.cos(0.27 * z + ph[1])
         + 0.35 * np.sin(0.9 * x + 0.6 * z + ph[2])
         + 0.12 * np.sin(2.3 * x + ph[3]) * np.sin(2.9 * z + ph[4])
         + 0.04 * np.sin(7.0 * (x - z) + ph[5]))
    verts = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(quads), np.arange(quads), indexing="ij")
    v00 = (i * (quads + 1) + j).reshape(-1)
    v01 = v00 + 1                # z + 1
    v10 = v00 + (quads + 1)      # x + 1
    v11 = v10 + 1
    tris = np.empty((quads * quads, 2, 3), np.uint32)
    tris[:, 0] = np.stack([v00, v01, v10], axis=-1)
    tris[:, 1] = np.stack([v10, v01, v11], axis=-1)
    s = Scene()
    solid = s.add_material("Terrain", abi.VRT_MAT_SOLID, (0.45, 0.7, 0.35))
    metal = s.add_material("Metal", abi.VRT_MAT_METAL, (0.9528, 0.9528, 0.9528))
    band = ((i // 59) + (j // 59)) % 3 == 0
    tm = np.where(band, metal, solid).astype(np.int32).reshape(-1)
    s.set_mesh(verts, tris.reshape(-1, 3), np.repeat(tm, 2))
    s.add_sphere((0.0, 3.0, 0.0), 1.5, metal)
    s.add_light(abi.VRT_LIGHT_POINT, (5.0, 25.0, -8.0), ls=1.0, cast_shadows=True)
    return s


def tessellated_mesh_camera(width=1920, height=1080):
    o = np.array([0.0, 9.0, -26.0])
    centre = np.array([0.0, 7.6, -23.0])
    fwd = centre - o
    fwd /= np.linalg.norm(fwd)
    right = np.cross([0.0, 1.0, 0.0], fwd)
    right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    hu = 2.0
    wu = hu * width / height
    p0 = centre - right * wu / 2 - up * hu / 2
    return make_camera(o, p0, right * wu, up * hu, width, height)
