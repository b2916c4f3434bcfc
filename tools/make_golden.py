#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ with the CPU oracle.

The reference ships no golden vectors (its test file asserts nothing), so these are outputs
of this repository's restatement of the reference algorithm, kept so that both the oracle and
the GPU path are pinned to a committed result.  Run from the repo root: python tools/make_golden.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402

import oracle  # noqa: E402
import raytracing_amd as ra  # noqa: E402
from raytracing_amd import abi  # noqa: E402


def cases():
    """name -> (scene, camera, Params).  Sampler tables come from the oracle's own generator."""
    out = {}
    offs = oracle.sampler_table(abi.VRT_SAMPLER_HAMMERSLEY, 50, 83, 0.01, 0.005, 1)
    out["c1_teaching_200x200_hammersley50_d10"] = (
        ra.teaching_scene(point_light=False), ra.teaching_camera(200, 200),
        ra.Params(spp=50, offsets=offs, seed=11, max_depth=10))
    out["c2_teaching_480x270_1spp_d5"] = (
        ra.teaching_scene(), ra.teaching_camera(480, 270), ra.Params(spp=1, max_depth=5))
    out["c3_spheres2000_192x108_1spp_d8"] = (
        ra.random_spheres_scene(n=2000, seed=5), ra.random_spheres_camera(192, 108),
        ra.Params(spp=1, max_depth=8, rng=200.0))
    out["c4_mesh48_160x90_1spp_d5"] = (
        ra.tessellated_mesh_scene(quads=48, seed=1), ra.tessellated_mesh_camera(160, 90),
        ra.Params(spp=1, max_depth=5, rng=100.0))
    jit = oracle.sampler_table(abi.VRT_SAMPLER_JITTERED, 16, 7, 2.4 * (16 / 9) / 96, 2.4 / 54, 3)
    out["c3_spheres1000_96x54_jittered16_d8"] = (
        ra.random_spheres_scene(n=1000, seed=8), ra.random_spheres_camera(96, 54),
        ra.Params(spp=16, offsets=jit, seed=2, max_depth=8, rng=200.0))
    out["textured_quad_96x64_2spp_d4_repeat3"] = (
        ra.textured_quad_scene(0, 3.0), ra.textured_quad_camera(96, 64), ra.Params(spp=2, max_depth=4))
    out["teaching_160x90_fixes_d6"] = (
        ra.teaching_scene(), ra.teaching_camera(160, 90),
        ra.Params(spp=1, max_depth=6,
                  flags=abi.VRT_FIX_DIELECTRIC | abi.VRT_FIX_PRIMARY_SKY | abi.VRT_FIX_AO_BASIS))
    out["matte_transparent_160x90_2spp_d6"] = (
        ra.teaching_scene_all_materials(), ra.teaching_camera(160, 90),
        ra.Params(spp=2, offsets=host_free_table(2), seed=3, max_depth=6))
    out["transparent_tir_96x54_d5"] = (
        ra.teaching_scene_all_materials(transparent_ray_dir=(0.0, 0.99, 0.14)), ra.teaching_camera(96, 54),
        ra.Params(spp=1, max_depth=5))
    return out


def host_free_table(spp):
    """A small fixed sampler table (no library needed): offsets on a diagonal, two sets."""
    import numpy as np
    t = np.zeros((2, spp, 2), np.float32)
    for k in range(2):
        for i in range(spp):
            t[k, i] = (0.003 * (i + 1) * (k + 1), -0.002 * (i + 1))
    return t


def main():
    oracle.build()
    gdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gdir, exist_ok=True)
    for name, (scene, cam, prm) in cases().items():
        if os.path.exists(os.path.join(gdir, name + ".npz")) and "--all" not in sys.argv:
            continue  # committed fixtures are only rewritten on request
        rgba, hit, rays = oracle.render(scene, cam, prm.c, n_threads=8)
        assert hit.max() < 32767
        np.savez_compressed(os.path.join(gdir, name + ".npz"), hit=hit.astype(np.int16),
                            rgba_even=rgba[::2, ::2].astype(np.float32), rays=rays.astype(np.uint64))
        print(name, "rays", rays.tolist(), "hit fraction", float((hit >= 0).mean()))


if __name__ == "__main__":
    main()
