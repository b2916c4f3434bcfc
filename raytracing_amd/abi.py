"""ctypes mirror of include/vrt.h (struct layouts and constants only, no code paths).

Kept free of any library loading so that both the product binding (native.py) and the
test-side oracle binding can share the data formats.
"""
import ctypes as C

VRT_ABI_VERSION = 5

VRT_OK = 0
VRT_ERR_INVALID = -1
VRT_ERR_NO_DEVICE = -2
VRT_ERR_HIP = -3
VRT_ERR_STATE = -4
VRT_ERR_CAPACITY = -5

VRT_MAT_SOLID, VRT_MAT_METAL, VRT_MAT_DIELECTRIC, VRT_MAT_MATTE, VRT_MAT_TRANSPARENT = 1, 2, 3, 4, 5
VRT_LIGHT_DIRECTIONAL, VRT_LIGHT_POINT = 0, 1
(VRT_SAMPLER_REGULAR, VRT_SAMPLER_RANDOM, VRT_SAMPLER_JITTERED, VRT_SAMPLER_NROOKS,
 VRT_SAMPLER_MULTIJITTERED, VRT_SAMPLER_HAMMERSLEY) = range(6)
VRT_FIX_PRIMARY_SKY, VRT_FIX_AO_BASIS, VRT_FIX_DIELECTRIC = 1, 2, 4
VRT_OUT_LINEAR_RGB = 1 << 8
VRT_OUT_RGBA8 = 1 << 9
VRT_MODE_AUTO, VRT_MODE_MEGAKERNEL, VRT_MODE_WAVEFRONT = 0, 1, 2
(VRT_RAY_PRECHECK, VRT_RAY_PRIMARY, VRT_RAY_REFLECT, VRT_RAY_TRANSMIT, VRT_RAY_SHADOW,
 VRT_RAY_AO) = range(6)
VRT_RAY_CLASSES = 6
RAY_CLASS_NAMES = ("precheck", "primary", "reflect", "transmit", "shadow", "ao")
VRT_MAX_LIGHTS = 8
VRT_MAX_DEPTH = 15
VRT_MAX_GPUS = 16
VRT_MAX_PARTS = 64
VRT_NOALLOC_MAX_OFFSETS = 65536


class vrt_sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("material", C.c_int32)]


class vrt_box(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("axis", C.c_float * 9), ("half", C.c_float * 3),
                ("material", C.c_int32)]


class vrt_material(C.Structure):
    _fields_ = [("type", C.c_int32), ("color", C.c_float * 4)]


class vrt_light(C.Structure):
    _fields_ = [("type", C.c_int32), ("vec", C.c_float * 3), ("ls", C.c_float),
                ("color", C.c_float * 4), ("cast_shadows", C.c_int32),
                ("location", C.c_float * 3)]


class vrt_shading(C.Structure):
    _fields_ = [("solid_ka", C.c_float), ("solid_kd", C.c_float), ("solid_ks", C.c_float),
                ("solid_exp", C.c_float),
                ("metal_ka", C.c_float), ("metal_kd", C.c_float), ("metal_ks", C.c_float),
                ("metal_exp", C.c_float), ("metal_kr", C.c_float),
                ("metal_cr", C.c_float * 4),
                ("glass_ka", C.c_float), ("glass_kd", C.c_float), ("glass_ks", C.c_float),
                ("glass_exp", C.c_float),
                ("glass_eta_in", C.c_float), ("glass_eta_out", C.c_float),
                ("ao_min_amount", C.c_float * 4), ("ao_ls", C.c_float),
                ("ao_color", C.c_float * 4),
                ("matte_ka", C.c_float), ("matte_kd", C.c_float),
                ("transp_ka", C.c_float), ("transp_kd", C.c_float), ("transp_ks", C.c_float),
                ("transp_exp", C.c_float),
                ("transp_ior", C.c_float), ("transp_kr", C.c_float), ("transp_kt", C.c_float),
                ("transp_ray_dir", C.c_float * 3)]


class vrt_texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("wrap", C.c_int32),
                ("rgba", C.POINTER(C.c_float))]


class vrt_scene_desc(C.Structure):
    _fields_ = [("n_spheres", C.c_uint32), ("spheres", C.POINTER(vrt_sphere)),
                ("n_boxes", C.c_uint32), ("boxes", C.POINTER(vrt_box)),
                ("n_vertices", C.c_uint32), ("vertices", C.POINTER(C.c_float)),
                ("n_triangles", C.c_uint32), ("indices", C.POINTER(C.c_uint32)),
                ("tri_material", C.POINTER(C.c_int32)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(vrt_material)),
                ("n_lights", C.c_uint32), ("lights", C.POINTER(vrt_light)),
                ("shading", vrt_shading),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(vrt_texture)),
                ("material_texture", C.POINTER(C.c_int32)), ("uvs", C.POINTER(C.c_float))]


class vrt_camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("p0", C.c_float * 3), ("x_axis", C.c_float * 3),
                ("y_axis", C.c_float * 3), ("width", C.c_int32), ("height", C.c_int32)]


class vrt_path_seg(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("end", C.c_float * 3), ("kind", C.c_int32),
                ("depth", C.c_int32), ("hit_prim", C.c_int32), ("sample", C.c_int32)]


class vrt_params(C.Structure):
    _fields_ = [("spp", C.c_int32), ("n_sets", C.c_int32), ("offsets", C.POINTER(C.c_float)),
                ("seed", C.c_uint32), ("max_depth", C.c_int32), ("range", C.c_float),
                ("flags", C.c_uint32), ("mode", C.c_int32),
                ("tile_index", C.c_int32), ("tile_count", C.c_int32), ("tile_rows", C.c_int32),
                ("compact_rows", C.c_int32),
                ("n_record_pixels", C.c_int32), ("record_pixels", C.POINTER(C.c_int32)),
                ("record_cap", C.c_int32), ("record_out", C.POINTER(vrt_path_seg)),
                ("record_count_out", C.POINTER(C.c_int32))]


class vrt_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64 * VRT_RAY_CLASSES), ("pixels", C.c_uint64),
                ("gpu_ms", C.c_float), ("bvh_build_ms", C.c_float),
                ("bvh_nodes", C.c_uint32), ("mode_used", C.c_int32),
                ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64)]


# Every symbol include/vrt.h declares; tests check the built library exports all of them.
EXPORTS = (
    "vrt_create", "vrt_destroy", "vrt_last_error", "vrt_shading_defaults",
    "vrt_params_defaults", "vrt_set_scene", "vrt_set_camera", "vrt_render",
    "vrt_render_begin", "vrt_render_poll", "vrt_render_end", "vrt_render_device", "vrt_render_device_noalloc",
    "vrt_fetch_stats", "vrt_tonemap_device", "vrt_trace_pixel", "vrt_trace_ray", "vrt_sampler_table",
    "vrt_pixel_set", "vrt_abi_version", "vrt_create_multi", "vrt_set_option", "vrt_host_register",
    "vrt_host_unregister", "vrt_host_alloc", "vrt_host_free", "vrt_render_begin_parts", "vrt_render_poll_parts",
)
