"""Render-parameter helper shared by the product binding and the tests."""
import ctypes as C

import numpy as np

from . import abi


class Params:
    """Owns the numpy buffers a vrt_params points at.

    Defaults are the shipped inspector values (Assets/Scenes/VisualRayTrace.unity:13859-13873
    and Assets/Scripts/RayTracer/Util/RayTraceUtility.cs:81,461): range 30, MaxDepth 10."""

    def __init__(self, spp=1, offsets=None, seed=0, max_depth=10, rng=30.0, flags=0,
                 mode=abi.VRT_MODE_AUTO, tile_index=0, tile_count=1, tile_rows=8,
                 compact_rows=False):
        self.c = abi.vrt_params()
        self.c.spp = spp
        self.offsets = None
        if offsets is not None:
            off = np.ascontiguousarray(offsets, np.float32)
            assert off.ndim == 3 and off.shape[1] == spp and off.shape[2] == 2, off.shape
            self.offsets = off
            self.c.n_sets = off.shape[0]
            self.c.offsets = off.ctypes.data_as(C.POINTER(C.c_float))
        else:
            self.c.n_sets = 1
        self.c.seed = seed
        self.c.max_depth = max_depth
        self.c.range = rng
        self.c.flags = flags
        self.c.mode = mode
        self.c.tile_index, self.c.tile_count, self.c.tile_rows = tile_index, tile_count, tile_rows
        self.c.compact_rows = int(compact_rows)
        self._rec = None

    def record(self, pixels, cap):
        """Ask vrt_render to record the ray tree of the given linear pixel indices."""
        px = np.ascontiguousarray(pixels, np.int32)
        segs = (abi.vrt_path_seg * (len(px) * cap))()
        counts = np.zeros(len(px), np.int32)
        self._rec = (px, segs, counts)
        self.c.n_record_pixels = len(px)
        self.c.record_pixels = px.ctypes.data_as(C.POINTER(C.c_int32))
        self.c.record_cap = cap
        self.c.record_out = segs
        self.c.record_count_out = counts.ctypes.data_as(C.POINTER(C.c_int32))
        return segs, counts
