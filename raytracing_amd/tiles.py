"""Image-strip sharding across ranks and the frame gather (one process per GPU).

The frame is cut into strips of `tile_rows` rows; strip s belongs to rank s % world.  Pixels
are independent, so the only exchange is the final gather of the strips to rank 0
(torch.distributed: RCCL over xGMI for GPU tensors, gloo for CPU tensors in the tests).
libvrt renders a rank's strips packed in ascending row order (vrt_params.compact_rows).
"""
import numpy as np
import torch
import torch.distributed as dist


def owned_rows(height, index, count, tile_rows):
    """Global row numbers rank `index` of `count` renders, ascending."""
    if count <= 1:
        return list(range(height))
    rows = []
    s = index
    while s * tile_rows < height:
        rows.extend(range(s * tile_rows, min((s + 1) * tile_rows, height)))
        s += count
    return rows


def max_owned_rows(height, count, tile_rows):
    return max(len(owned_rows(height, k, count, tile_rows)) for k in range(max(count, 1)))


class FrameGather:
    """Gathers packed strips to rank 0 and puts them back in row order.

    Every rank passes a buffer of `rows_max` rows (ranks that own fewer leave the tail unused)
    so that the gather moves equal-sized messages; rank 0 then scatters the rows into the full
    frame with one index_copy (unused tail rows land in a spare row)."""

    def __init__(self, height, width, channels, world, rank, tile_rows, device, dtype=torch.float32,
                 group=None):
        self.h, self.w, self.c = height, width, channels
        self.world, self.rank, self.tile_rows, self.group = world, rank, tile_rows, group
        self.rows_max = max_owned_rows(height, world, tile_rows)
        self.local = torch.zeros((self.rows_max, width, channels), dtype=dtype, device=device)
        self.n_local = len(owned_rows(height, rank, world, tile_rows))
        self.gathered = self.final = self._final_buf = self.dst_index = None
        if rank == 0:
            # one spare row at the end takes the unused tail rows of ranks that own fewer rows, so
            # that the de-interleave is a single index_copy over everything received
            self._final_buf = torch.zeros((height + 1, width, channels), dtype=dtype, device=device)
            self.final = self._final_buf[:height]
            if world > 1:
                self.gathered = torch.zeros((world, self.rows_max, width, channels), dtype=dtype, device=device)
                dst = []
                for k in range(world):
                    rows = owned_rows(height, k, world, tile_rows)
                    dst.extend(rows)
                    dst.extend([height] * (self.rows_max - len(rows)))
                self.dst_index = torch.tensor(dst, dtype=torch.long, device=device)

    def gather(self):
        """Collective: call on every rank after `local` has been filled.  Returns the full
        frame on rank 0, None elsewhere."""
        if self.world <= 1:
            self.final[: self.n_local].copy_(self.local[: self.n_local])
            return self.final
        if self.rank == 0:
            dist.gather(self.local, list(self.gathered.unbind(0)), dst=0, group=self.group)
            self._deinterleave()
            return self.final
        dist.gather(self.local, None, dst=0, group=self.group)
        return None

    # ---- pipelined form: the gather of frame k runs on the collective's own stream while the
    # caller renders frame k+1 into another FrameGather's buffer.

    def gather_async(self):
        """Starts the gather of `local` (after the work already enqueued on the current stream)
        and returns a handle for finish().  With world == 1 there is nothing to move."""
        if self.world <= 1:
            return None
        outs = list(self.gathered.unbind(0)) if self.rank == 0 else None
        return dist.gather(self.local, outs, dst=0, group=self.group, async_op=True)

    def finish(self, handle):
        """Makes the current stream wait for the gather and, on rank 0, puts the strips in row
        order.  After it returns `local` may be overwritten by work on the current stream."""
        if self.world <= 1:
            return self.final
        handle.wait()
        if self.rank == 0:
            self._deinterleave()
            return self.final
        return None

    def deinterleave(self):
        """Rank 0, after the gather has completed on the current stream: strips -> row order.
        For callers that wait for the gather themselves (e.g. on a stream of their own)."""
        self._deinterleave()

    def _deinterleave(self):
        flat = self.gathered.view(self.world * self.rows_max, self.w, self.c)
        self._final_buf.index_copy_(0, self.dst_index, flat)


# ---- scene replication: rank 0 owns the scene description (in the Unity host: the process that
# marshalled the colliders), every other rank receives a byte-identical copy before it uploads it
# to its own GPU.  One broadcast of a small header with the array sizes, then one per array.

_SCENE_ARRAYS = ("spheres", "boxes", "vertices", "indices", "tri_material", "materials", "lights", "uvs")


def broadcast_scene(scene, device, src=0, group=None):
    """Collective.  Rank `src` passes its Scene, the others pass a fresh Scene(); returns the
    scene every rank should upload.  `device` is where the transfer buffers live (the rank's GPU
    for RCCL, CPU for gloo)."""
    rank = dist.get_rank(group)
    n_tex = torch.zeros(1, dtype=torch.int64, device=device)
    header = torch.zeros(len(_SCENE_ARRAYS), dtype=torch.int64, device=device)
    if rank == src:
        sizes = [(-1 if getattr(scene, k) is None else np.ascontiguousarray(getattr(scene, k)).nbytes)
                 for k in _SCENE_ARRAYS]
        header.copy_(torch.tensor(sizes, dtype=torch.int64))
        n_tex[0] = len(scene.textures)
    dist.broadcast(header, src, group=group)
    dist.broadcast(n_tex, src, group=group)
    proto = type(scene)()
    for k, nbytes in zip(_SCENE_ARRAYS, header.tolist()):
        if nbytes < 0:
            setattr(scene, k, None)
            continue
        like = getattr(proto, k)
        dtype = {"tri_material": np.dtype(np.int32), "uvs": np.dtype(np.float32)}.get(k, None if like is None else like.dtype)
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        if rank == src and nbytes:
            buf[:nbytes].copy_(torch.from_numpy(np.ascontiguousarray(getattr(scene, k)).view(np.uint8).reshape(-1).copy()))
        dist.broadcast(buf, src, group=group)
        if rank != src:
            arr = np.frombuffer(buf[:nbytes].cpu().numpy().tobytes(), dtype=dtype).copy()
            tail = {"vertices": (-1, 3), "indices": (-1, 3), "uvs": (-1, 2)}.get(k)
            setattr(scene, k, arr.reshape(tail) if tail else arr)
    # shading constants, material textures and textures
    sh_bytes = bytes(scene.shading) if rank == src else bytes(len(bytes(scene.shading)))
    sh = torch.tensor(list(sh_bytes), dtype=torch.uint8, device=device)
    dist.broadcast(sh, src, group=group)
    if rank != src:
        scene.shading = type(scene.shading).from_buffer_copy(bytes(sh.cpu().tolist()))
    mt = torch.full((max(len(scene.materials), 1),), -1, dtype=torch.int64, device=device)
    if rank == src and scene.mat_texture:
        vals = list(scene.mat_texture) + [-1] * (len(scene.materials) - len(scene.mat_texture))
        mt[: len(scene.materials)].copy_(torch.tensor(vals[: len(scene.materials)], dtype=torch.int64))
    dist.broadcast(mt, src, group=group)
    if rank != src:
        scene.mat_texture = mt[: len(scene.materials)].tolist()
        scene.textures = []
    for t in range(int(n_tex[0])):
        meta = torch.zeros(3, dtype=torch.int64, device=device)
        if rank == src:
            img, wrap = scene.textures[t]
            meta.copy_(torch.tensor([img.shape[0], img.shape[1], wrap], dtype=torch.int64))
        dist.broadcast(meta, src, group=group)
        h, w, wrap = meta.tolist()
        tex = torch.empty((h, w, 4), dtype=torch.float32, device=device)
        if rank == src:
            tex.copy_(torch.from_numpy(scene.textures[t][0]))
        dist.broadcast(tex, src, group=group)
        if rank != src:
            scene.textures.append((tex.cpu().numpy().copy(), int(wrap)))
    return scene
