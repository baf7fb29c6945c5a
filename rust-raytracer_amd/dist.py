"""Multi-GPU frame sharding: interleaved scanline tiles + ONE gather at frame end.

The reference parallelises over scanlines inside one process (rayon, raytracer.rs:255-262).
Across GPUs the same decomposition is used: tile k (TILE_ROWS scanlines) belongs to rank
k mod G — finely interleaved because row cost is very non-uniform (sky rows are 1 segment/sample,
ground rows 3+).  Pixels are independent and the RNG is addressed by GLOBAL pixel index, so
the assembled frame is bit-identical for every G.  There is no intra-frame communication;
the only collective is one `gather` of the packed RGB8 tiles to rank 0 (RCCL over xGMI when
the backend is "nccl", gloo in the CPU tests).  The renderer is a parameter: this module
never imports a renderer itself.
"""
import torch
import torch.distributed as dist

from . import abi

TILE_ROWS = 2  # measured: 2-row interleave balances the ranks to +-2 % (8 rows: +-6 %), profiles/r01_run4_shards.log


def shard(rank, world, tile_rows=TILE_ROWS):
    """RtRowTiles for `rank` of `world` (None = whole frame when world == 1)."""
    if world <= 1:
        return None
    return abi.RtRowTiles(tile_rows, rank, world)


def max_local_rows(height, world, tile_rows=TILE_ROWS):
    return max(abi.tiles_local_rows(height, shard(r, world, tile_rows)) for r in range(world))


_PERM_CACHE = {}


def _assembly_perm(height, world, tile_rows, pad_rows, device):
    """perm[y] = rank * pad_rows + local_row of global scanline y (built once per layout)."""
    key = (height, world, tile_rows, pad_rows, str(device))
    perm = _PERM_CACHE.get(key)
    if perm is None:
        idx = [0] * height
        for r in range(world):
            for lr, y in enumerate(abi.tiles_global_rows(height, shard(r, world, tile_rows))):
                idx[y] = r * pad_rows + lr
        perm = torch.as_tensor(idx, device=device, dtype=torch.long)
        _PERM_CACHE[key] = perm
    return perm


def gather_frame(local_rgb8, height, width, rank, world, tile_rows=TILE_ROWS, dst=0, group=None):
    """local_rgb8: uint8 tensor [max_local_rows, width, 3] on this rank's device (rows beyond
    this rank's share are padding).  Returns the assembled [height, width, 3] frame on `dst`,
    None elsewhere.  ONE collective (gather of the packed tiles) + one row permutation on dst."""
    if world <= 1:
        return local_rgb8[:height]
    pad_rows = max_local_rows(height, world, tile_rows)
    assert local_rgb8.shape[0] == pad_rows, (local_rgb8.shape, pad_rows)
    if rank == dst:
        stacked = torch.empty((world, pad_rows, width, 3), dtype=torch.uint8, device=local_rgb8.device)
        dist.gather(local_rgb8, [stacked[r] for r in range(world)], dst=dst, group=group)
        perm = _assembly_perm(height, world, tile_rows, pad_rows, local_rgb8.device)
        return stacked.view(world * pad_rows, width, 3).index_select(0, perm)  # de-interleave: packed rows -> scanlines
    dist.gather(local_rgb8, None, dst=dst, group=group)
    return None


class FramePipeline:
    """Double-buffered frames for a stream of renders (bench.py, animation): frame i's gather runs
    asynchronously (on the collective's own stream with RCCL) while frame i+1 is being rendered
    into the other buffer.  Per frame still ONE collective + one row permutation on `dst`.

        buf, done = pipe.begin(i)     # tile buffer to render frame i into; `done` = frame i-depth
        ... render into buf on the current stream ...
        pipe.submit(i)                # gather of frame i starts once the render has finished
        frames = pipe.drain()         # after the last frame: the frames still in flight, in order

    `group`: the process group the gather runs on (None = the default one).  `host_staged`: the fall-back for a node whose
    RCCL does not come up (bench.py decides, all ranks together): the tiles leave each GPU into pinned host memory, the
    gather runs over a CPU group (gloo) and the frame is assembled in host memory of `dst` — same bytes, no xGMI.
    """

    def __init__(self, height, width, rank, world, device, tile_rows=TILE_ROWS, dst=0, depth=2, force_collective=False, group=None, host_staged=False):
        self.h, self.w, self.rank, self.world, self.dst, self.depth, self.tile_rows = height, width, rank, world, dst, depth, tile_rows
        self.collective = world > 1 or force_collective   # (force_collective: a 1-rank group still goes through the gather; tests)
        self.group, self.host_staged = group, bool(host_staged and self.collective)
        self.pad_rows = max_local_rows(height, world, tile_rows) if world > 1 else height
        self.local = [torch.zeros((self.pad_rows, width, 3), dtype=torch.uint8, device=device) for _ in range(depth)]
        gdev = "cpu" if self.host_staged else device
        self.host_local = [torch.zeros((self.pad_rows, width, 3), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
                           for _ in range(depth)] if self.host_staged else None
        self.stacked = [torch.empty((world, self.pad_rows, width, 3), dtype=torch.uint8, device=gdev)
                        if (self.collective and rank == dst) else None for _ in range(depth)]
        self.work = [None] * depth
        self.order = []  # buffer slots with a gather in flight, oldest first

    def _collect(self, slot):
        work, self.work[slot] = self.work[slot], None
        if not self.collective:
            return self.local[slot][: self.h]
        if work is not None:
            work.wait()  # the current stream (or the host, with gloo) waits for the collective
        if self.rank != self.dst:
            return None
        perm = _assembly_perm(self.h, self.world, self.tile_rows, self.pad_rows, self.stacked[slot].device)
        return self.stacked[slot].view(self.world * self.pad_rows, self.w, 3).index_select(0, perm)

    def begin(self, i):
        slot = i % self.depth
        done = None
        if slot in self.order:  # the buffer is still owned by frame i - depth: finish that one first
            self.order.remove(slot)
            done = self._collect(slot)
        return self.local[slot], done

    def submit(self, i):
        slot = i % self.depth
        if self.collective:
            outs = [self.stacked[slot][r] for r in range(self.world)] if self.rank == self.dst else None
            src = self.local[slot]
            if self.host_staged:   # device -> pinned host, then the CPU gather (it runs on gloo's own thread, under the next frame's render)
                src = self.host_local[slot]
                if self.local[slot].is_cuda:
                    src.copy_(self.local[slot], non_blocking=True)
                    torch.cuda.current_stream().synchronize()
                else:
                    src.copy_(self.local[slot])
            self.work[slot] = dist.gather(src, outs, dst=self.dst, async_op=True, group=self.group)
        self.order.append(slot)

    def drain(self):
        frames = [self._collect(slot) for slot in self.order]
        self.order = []
        return frames
