"""ctypes binding of the CPU oracle (oracle/vrt_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracing_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvrt_oracle.so")
_lib = None


def build(force=False):
    """make decides what is out of date (vrt_oracle.c and include/vrt.h are its prerequisites)."""
    if os.path.exists(os.path.join(_HERE, "Makefile")) and os.path.exists(os.path.join(_HERE, "vrt_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.vrto_render.restype = C.c_int
        L.vrto_render.argtypes = [C.POINTER(abi.vrt_scene_desc), C.POINTER(abi.vrt_camera),
                                  C.POINTER(abi.vrt_params), C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int]
        L.vrto_trace_pixel.restype = C.c_int
        L.vrto_trace_pixel.argtypes = [C.POINTER(abi.vrt_scene_desc), C.POINTER(abi.vrt_camera),
                                       C.POINTER(abi.vrt_params), C.c_int, C.c_int, C.c_int,
                                       C.POINTER(abi.vrt_path_seg), C.c_int,
                                       C.POINTER(C.c_int), C.POINTER(C.c_float),
                                       C.POINTER(C.c_int32), C.c_void_p]
        L.vrto_intersect.restype = C.c_int
        L.vrto_intersect.argtypes = [C.POINTER(abi.vrt_scene_desc), C.POINTER(C.c_float),
                                     C.POINTER(C.c_float), C.c_float, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.vrto_sampler_table.restype = C.c_int
        L.vrto_sampler_table.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                         C.c_uint64, C.c_void_p]
        L.vrto_pixel_set.restype = C.c_uint32
        L.vrto_pixel_set.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.vrto_ao_direction.restype = C.c_int
        _lib = L
    return _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def render(scene, cam, params, n_threads=1):
    """Returns (rgba[H,W,4] float32, hit_id[H,W] int32, rays[6] uint64)."""
    d = scene.desc()
    H, W = cam.height, cam.width
    rgba = np.zeros((H, W, 4), np.float32)
    hit = np.full((H, W), -1, np.int32)
    rays = np.zeros(abi.VRT_RAY_CLASSES, np.uint64)
    rc = lib().vrto_render(C.byref(d), C.byref(cam), C.byref(params), rgba.ctypes.data,
                           hit.ctypes.data, rays.ctypes.data, n_threads)
    assert rc == 0
    return rgba, hit, rays


def trace_pixel(scene, cam, params, h, v, sample=-1, cap=256):
    d = scene.desc()
    segs = (abi.vrt_path_seg * max(cap, 1))()
    n = C.c_int(0)
    rgba = (C.c_float * 4)()
    hit = C.c_int32(-1)
    rays = np.zeros(abi.VRT_RAY_CLASSES, np.uint64)
    lib().vrto_trace_pixel(C.byref(d), C.byref(cam), C.byref(params), h, v, sample,
                           segs if cap > 0 else None, cap, C.byref(n), rgba, C.byref(hit),
                           rays.ctypes.data)
    return np.array(rgba[:], np.float32), hit.value, segs, n.value, rays


def trace_ray(scene, params, o, dirv, depth=0):
    d = scene.desc()
    rgba = (C.c_float * 4)()
    hit = C.c_int32(-1)
    rays = np.zeros(abi.VRT_RAY_CLASSES, np.uint64)
    lib().vrto_trace_ray(C.byref(d), C.byref(params), _f3(o), _f3(dirv), depth, rgba, C.byref(hit),
                         C.c_void_p(rays.ctypes.data))
    return np.array(rgba[:], np.float32), hit.value, rays


def intersect(scene, o, dirv, maxd=30.0, skip=-1, any_hit=False, force_brute=False):
    d = scene.desc()
    t = C.c_float(0)
    prim = C.c_int32(-1)
    pt = (C.c_float * 3)()
    nr = (C.c_float * 3)()
    hit = lib().vrto_intersect(C.byref(d), _f3(o), _f3(dirv), maxd, skip, int(any_hit),
                               int(force_brute), C.byref(t), C.byref(prim), pt, nr)
    return bool(hit), t.value, prim.value, np.array(pt[:], np.float32), np.array(nr[:], np.float32)


def sampler_table(sampler, n, sets, h_step, v_step, seed):
    out = np.zeros((sets, n, 2), np.float32)
    rc = lib().vrto_sampler_table(sampler, n, sets, h_step, v_step, seed, out.ctypes.data)
    assert rc == 0
    return out


def sky_color(d):
    out = (C.c_float * 4)()
    lib().vrto_sky_color(_f3(d), out)
    return np.array(out[:], np.float32)


def process_color(rgb):
    out = (C.c_float * 4)()
    lib().vrto_process_color(_f3(rgb), out)
    return np.array(out[:], np.float32)


def camera_dir(cam, h, v):
    out = (C.c_float * 3)()
    lib().vrto_camera_dir(C.byref(cam), h, v, out)
    return np.array(out[:], np.float32)


def ao_direction(n):
    out = (C.c_float * 3)()
    ok = lib().vrto_ao_direction(_f3(n), out)
    return bool(ok), np.array(out[:], np.float32)


def pixel_set(seed, pixel, n_sets):
    return lib().vrto_pixel_set(seed, pixel, n_sets)


def trace_pixels(scene, cam, params, pixels):
    """Oracle result for a list of linear pixel indices: (rgba[n,4], hit[n], rays[6])."""
    d = scene.desc()
    px = np.ascontiguousarray(pixels, np.int32)
    rgba = np.zeros((len(px), 4), np.float32)
    hit = np.zeros(len(px), np.int32)
    rays = np.zeros(abi.VRT_RAY_CLASSES, np.uint64)
    lib().vrto_trace_pixels(C.byref(d), C.byref(cam), C.byref(params), len(px),
                            C.c_void_p(px.ctypes.data), C.c_void_p(rgba.ctypes.data),
                            C.c_void_p(hit.ctypes.data), C.c_void_p(rays.ctypes.data))
    return rgba, hit, rays


def rgba8(rgba):
    """The Color32 an RGBA32 Texture2D holds after SetPixels(Color[]) + Apply
    (Assets/Scripts/RayTracer/RayTracerUnity.cs:472-475; UnityEngine's Color -> Color32 conversion is
    round(clamp01(c) * 255) per channel in float32, halves to even): the frame as the player shows it.
    NaN channels (possible in the float frame, see the ambient occluder) become 0."""
    x = np.nan_to_num(np.asarray(rgba, np.float32), nan=0.0, posinf=1.0, neginf=0.0)
    return np.rint(np.clip(x, np.float32(0), np.float32(1)) * np.float32(255)).astype(np.uint8)
