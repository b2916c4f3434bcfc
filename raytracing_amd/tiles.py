"""Image-strip sharding across ranks and the frame gather (one process per GPU).

The frame is cut into strips of `tile_rows` rows; strip s belongs to rank s % world.  Pixels
are independent, so the only exchange is the final gather of the strips to rank 0
(torch.distributed: RCCL over xGMI for GPU tensors, gloo for CPU tensors in the tests).
libvrt renders a rank's strips packed in ascending row order (vrt_params.compact_rows).
"""
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

    def _deinterleave(self):
        flat = self.gathered.view(self.world * self.rows_max, self.w, self.c)
        self._final_buf.index_copy_(0, self.dst_index, flat)
