"""ctypes binding of libvrt.so — the only route from Python to the HIP kernels.

There is no CPU fallback: if the library is missing or no GPU is present the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import abi

# VRT_LIB points at an alternative build of the same library (kernel A/B experiments).
_LIB_PATH = os.environ.get("VRT_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvrt.so")
_lib = None


class VrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvrt error {code}: {msg}")
        self.code = code


def library_path():
    return _LIB_PATH


def load_library():
    """Loads raytracing_amd/libvrt.so (built by `make -C raytracing_amd/csrc` or
    __graft_entry__.build()).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(f"{_LIB_PATH} is missing: build it with __graft_entry__.build(); "
                           "there is no CPU fallback")
    L = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    L.vrt_abi_version.restype = C.c_int
    L.vrt_create.restype = C.c_int
    L.vrt_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.vrt_create_multi.restype = C.c_int
    L.vrt_create_multi.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    L.vrt_set_option.restype = C.c_int
    L.vrt_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
    L.vrt_host_alloc.restype = C.c_int
    L.vrt_host_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.vrt_host_free.restype = C.c_int
    L.vrt_host_free.argtypes = [vp, vp]
    L.vrt_host_register.restype = C.c_int
    L.vrt_host_register.argtypes = [vp, vp, C.c_uint64]
    L.vrt_host_unregister.restype = C.c_int
    L.vrt_host_unregister.argtypes = [vp, vp]
    L.vrt_render_begin_parts.restype = C.c_int
    L.vrt_render_begin_parts.argtypes = [vp, C.POINTER(abi.vrt_params), vp, vp, C.c_int]
    L.vrt_render_poll_parts.restype = C.c_int
    L.vrt_render_poll_parts.argtypes = [vp, C.POINTER(C.c_int)]
    L.vrt_destroy.restype = C.c_int
    L.vrt_destroy.argtypes = [vp]
    L.vrt_last_error.restype = C.c_char_p
    L.vrt_last_error.argtypes = [vp]
    L.vrt_shading_defaults.restype = None
    L.vrt_shading_defaults.argtypes = [C.POINTER(abi.vrt_shading)]
    L.vrt_params_defaults.restype = None
    L.vrt_params_defaults.argtypes = [C.POINTER(abi.vrt_params)]
    L.vrt_set_scene.restype = C.c_int
    L.vrt_set_scene.argtypes = [vp, C.POINTER(abi.vrt_scene_desc)]
    L.vrt_set_camera.restype = C.c_int
    L.vrt_set_camera.argtypes = [vp, C.POINTER(abi.vrt_camera)]
    L.vrt_render.restype = C.c_int
    L.vrt_render.argtypes = [vp, C.POINTER(abi.vrt_params), vp, vp, C.POINTER(abi.vrt_stats)]
    L.vrt_render_begin.restype = C.c_int
    L.vrt_render_begin.argtypes = [vp, C.POINTER(abi.vrt_params), vp, vp]
    L.vrt_render_poll.restype = C.c_int
    L.vrt_render_poll.argtypes = [vp, C.POINTER(C.c_int)]
    L.vrt_render_end.restype = C.c_int
    L.vrt_render_end.argtypes = [vp, C.POINTER(abi.vrt_stats)]
    L.vrt_render_device.restype = C.c_int
    L.vrt_render_device.argtypes = [vp, C.POINTER(abi.vrt_params), vp, vp, vp]
    L.vrt_render_device_noalloc.restype = C.c_int
    L.vrt_render_device_noalloc.argtypes = [vp, C.POINTER(abi.vrt_params), vp, vp, vp]
    L.vrt_tonemap_device.restype = C.c_int
    L.vrt_tonemap_device.argtypes = [vp, vp, vp, C.c_uint64, vp]
    L.vrt_fetch_stats.restype = C.c_int
    L.vrt_fetch_stats.argtypes = [vp, C.POINTER(abi.vrt_stats)]
    L.vrt_trace_pixel.restype = C.c_int
    L.vrt_trace_pixel.argtypes = [vp, C.POINTER(abi.vrt_params), C.c_int, C.c_int, C.c_int,
                                  C.POINTER(abi.vrt_path_seg), C.c_int, C.POINTER(C.c_int),
                                  C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.vrt_trace_ray.restype = C.c_int
    L.vrt_trace_ray.argtypes = [vp, C.POINTER(abi.vrt_params), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.vrt_sampler_table.restype = C.c_int
    L.vrt_sampler_table.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint64, vp]
    L.vrt_pixel_set.restype = C.c_uint32
    L.vrt_pixel_set.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    if L.vrt_abi_version() != abi.VRT_ABI_VERSION:
        raise RuntimeError("libvrt.so ABI version mismatch; rebuild it")
    _lib = L
    return L


def sampler_table(sampler, n_samples, n_sets, h_step, v_step, seed=0):
    """AntiAliasingStrategy's sampler table in access order, shape (n_sets, n_samples, 2)."""
    out = np.zeros((n_sets, n_samples, 2), np.float32)
    rc = load_library().vrt_sampler_table(sampler, n_samples, n_sets, h_step, v_step, seed,
                                          out.ctypes.data)
    if rc != 0:
        raise VrtError(rc, "vrt_sampler_table: invalid arguments")
    return out


def pixel_set(seed, pixel, n_sets):
    return load_library().vrt_pixel_set(seed, pixel, n_sets)


class Context:
    """One vrt_ctx: on one GPU (device=ordinal) or on several GPUs of the node (devices=list of
    ordinals, vrt_create_multi)."""

    def __init__(self, device=0, devices=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            rc = self._lib.vrt_create_multi(len(devices), arr, C.byref(self._h))
            device = devices[0] if len(devices) else 0
        else:
            rc = self._lib.vrt_create(device, C.byref(self._h))
        if rc != 0:
            self._h = None
            raise VrtError(rc, "vrt_create failed (no usable HIP device?)")
        self.device = device
        self.devices = list(devices) if devices is not None else None
        self._scene = None
        self.cam = None

    def close(self):
        if self._h:
            self._lib.vrt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise VrtError(rc, self._lib.vrt_last_error(self._h).decode())

    def set_option(self, name, value):
        """Study/test knob by name (vrt_set_option); none changes a result."""
        self._chk(self._lib.vrt_set_option(self._h, name.encode(), int(value)))

    def host_alloc(self, shape, dtype=np.float32):
        """A numpy array over page-locked memory of the library (vrt_host_alloc); frames land in it
        at PCIe speed.  Valid until host_free(array) or close()."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._chk(self._lib.vrt_host_alloc(self._h, n, C.byref(p)))
        buf = (C.c_char * n).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def host_free(self, array):
        self._chk(self._lib.vrt_host_free(self._h, array.ctypes.data))

    def host_register(self, array):
        """Page-locks a numpy array that will receive frames (vrt_host_register)."""
        self._chk(self._lib.vrt_host_register(self._h, array.ctypes.data, array.nbytes))

    def host_unregister(self, array):
        self._chk(self._lib.vrt_host_unregister(self._h, array.ctypes.data))

    def set_scene(self, scene):
        d = scene.desc()
        self._chk(self._lib.vrt_set_scene(self._h, C.byref(d)))
        self._scene = scene

    def set_camera(self, cam):
        self._chk(self._lib.vrt_set_camera(self._h, C.byref(cam)))
        self.cam = cam

    def render(self, params, rgba=None, hit=None, want_hit=True):
        """Whole frame into host arrays. Returns (rgba[H,W,4], hit[H,W] or None, vrt_stats)."""
        H, W = self.cam.height, self.cam.width
        if rgba is None:   # VRT_OUT_RGBA8: a Color32 per pixel instead of a Color
            rgba = np.zeros((H, W, 4), np.uint8 if params.c.flags & abi.VRT_OUT_RGBA8 else np.float32)
        if hit is None and want_hit:
            hit = np.full((H, W), -1, np.int32)
        st = abi.vrt_stats()
        self._chk(self._lib.vrt_render(self._h, C.byref(params.c), rgba.ctypes.data,
                                       hit.ctypes.data if hit is not None else None, C.byref(st)))
        return rgba, hit, st

    def render_begin(self, params, rgba, hit=None):
        """Starts a frame into the given host arrays (keep them alive until render_end)."""
        self._chk(self._lib.vrt_render_begin(self._h, C.byref(params.c), rgba.ctypes.data,
                                             hit.ctypes.data if hit is not None else None))

    def render_begin_parts(self, params, rgba, hit=None, n_parts=4):
        """Starts a frame delivered in n_parts strip sets (keep the arrays alive until render_end)."""
        self._chk(self._lib.vrt_render_begin_parts(self._h, C.byref(params.c), rgba.ctypes.data,
                                                   hit.ctypes.data if hit is not None else None, n_parts))

    def render_poll_parts(self):
        n = C.c_int(0)
        self._chk(self._lib.vrt_render_poll_parts(self._h, C.byref(n)))
        return n.value

    def render_poll(self):
        done = C.c_int(0)
        self._chk(self._lib.vrt_render_poll(self._h, C.byref(done)))
        return bool(done.value)

    def render_end(self):
        st = abi.vrt_stats()
        self._chk(self._lib.vrt_render_end(self._h, C.byref(st)))
        return st

    def render_device(self, params, rgba_ptr, hit_ptr=None, stream=None):
        """Enqueues a frame whose outputs stay in device memory (raw device pointers)."""
        self._chk(self._lib.vrt_render_device(self._h, C.byref(params.c), rgba_ptr, hit_ptr,
                                              stream))

    def render_device_noalloc(self, params, rgba_ptr, hit_ptr=None, stream=None):
        """render_device for callers that cannot allocate: counters and sampler table in the
        library's static device array, nothing allocated or freed by the call."""
        self._chk(self._lib.vrt_render_device_noalloc(self._h, C.byref(params.c), rgba_ptr, hit_ptr,
                                                      stream))

    def tonemap_device(self, rgb_ptr, rgba_ptr, n_pixels, stream=None):
        """3 floats per pixel as VRT_OUT_LINEAR_RGB leaves them -> the frame's 4 floats per pixel."""
        self._chk(self._lib.vrt_tonemap_device(self._h, rgb_ptr, rgba_ptr, n_pixels, stream))

    def fetch_stats(self):
        st = abi.vrt_stats()
        self._chk(self._lib.vrt_fetch_stats(self._h, C.byref(st)))
        return st

    def trace_pixel(self, params, h, v, sample=-1, cap=256):
        segs = (abi.vrt_path_seg * max(cap, 1))()
        n = C.c_int(0)
        rgba = (C.c_float * 4)()
        hit = C.c_int32(-1)
        self._chk(self._lib.vrt_trace_pixel(self._h, C.byref(params.c), h, v, sample,
                                            segs if cap > 0 else None, cap, C.byref(n), rgba,
                                            C.byref(hit)))
        return np.array(rgba[:], np.float32), hit.value, segs, n.value

    def trace_ray(self, params, origin, direction, depth=0):
        """AbstractTracer.TraceRay(ray, depth): raw radiance of one ray and its first hit id."""
        o = (C.c_float * 3)(*[float(x) for x in origin])
        d = (C.c_float * 3)(*[float(x) for x in direction])
        rgba = (C.c_float * 4)()
        hit = C.c_int32(-1)
        self._chk(self._lib.vrt_trace_ray(self._h, C.byref(params.c), o, d, depth, rgba, C.byref(hit)))
        return np.array(rgba[:], np.float32), hit.value
