"""raytracing_amd — MI355X-native tracing core for VisualRayTrace's per-pixel loop.

The compute lives in raytracing_amd/csrc (HIP kernels behind the C ABI of include/vrt.h,
built into raytracing_amd/libvrt.so).  This package is the thin host-side mirror of the
reference's tracer-facing interface plus scene builders for the benchmark workloads.
Importing it never loads the native library; raytracing_amd.host does, and fails loudly
if it is missing.
"""
from . import abi  # noqa: F401
from .params import Params  # noqa: F401
from .scenes import (Scene, make_camera, teaching_scene, teaching_camera, teaching_scene_all_materials,  # noqa: F401
                     random_spheres_scene, random_spheres_camera,
                     tessellated_mesh_scene, tessellated_mesh_camera,
                     textured_quad_scene, textured_quad_camera, checker_texture)
