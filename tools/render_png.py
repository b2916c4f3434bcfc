#!/usr/bin/env python3
"""Renders a workload through libvrt and writes a PNG (needs a GPU).

    python tools/render_png.py teaching|spheres|mesh out.png [width height]
"""
import struct
import sys
import zlib
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import raytracing_amd as ra  # noqa: E402
from raytracing_amd import host  # noqa: E402


def write_png(path, img):
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[i].tobytes() for i in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def main():
    which, out = sys.argv[1], sys.argv[2]
    w = int(sys.argv[3]) if len(sys.argv) > 3 else 640
    h = int(sys.argv[4]) if len(sys.argv) > 4 else 360
    if which == "teaching":
        scene, cam, prm = ra.teaching_scene(), ra.teaching_camera(w, h), ra.Params(spp=1, max_depth=5)
    elif which == "spheres":
        scene, cam = ra.random_spheres_scene(10000), ra.random_spheres_camera(w, h)
        prm = ra.Params(spp=1, max_depth=8, rng=200.0)
    else:
        scene, cam = ra.tessellated_mesh_scene(708), ra.tessellated_mesh_camera(w, h)
        prm = ra.Params(spp=1, max_depth=5, rng=100.0)
    ctx = host.Context(0)
    ctx.set_scene(scene)
    ctx.set_camera(cam)
    rgba, hit, st = ctx.render(prm)
    print(which, "rays", list(st.rays), "bvh ms", st.bvh_build_ms, "gpu ms", st.gpu_ms)
    img = (np.clip(np.nan_to_num(rgba[::-1, :, :3]), 0, 1) * 255).astype(np.uint8)
    write_png(out, img)


if __name__ == "__main__":
    main()
