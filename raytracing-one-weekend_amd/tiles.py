"""Image partition + framebuffer gather for the multi-GPU path (one process per GPU).

The image is cut into horizontal strips of `tile_rows` rows dealt round-robin: strip t
belongs to rank t % nranks (include/rtow.h, rtow_config_t).  Every rank renders its rows
into a [max_rows, W, 3] float64 tensor; ONE torch.distributed.gather brings the strips to
rank 0 (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests), which
scatters them back to their global rows.  There is no other data-path collective: the
path shards by pixels, and the counter-based RNG makes a pixel independent of who traces
it (reference analogue: the in-order sum of per-thread images, src/render.cpp:176-180).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def strip_rows(height: int, tile_rows: int, nranks: int, rank: int):
    """Global row numbers owned by `rank`, ascending (same rule as rtow_local_row_list)."""
    return [i for i in range(height) if (i // tile_rows) % nranks == rank]


def max_rows(height: int, tile_rows: int, nranks: int) -> int:
    return max(len(strip_rows(height, tile_rows, nranks, r)) for r in range(nranks))


class StripGather:
    """Reusable buffers for gathering per-rank strips into the full image on rank 0."""

    def __init__(self, height, width, tile_rows, rank, world, device, dst=0):
        self.h, self.w, self.tile, self.rank, self.world, self.dst = height, width, tile_rows, rank, world, dst
        self.rows = strip_rows(height, tile_rows, world, rank)
        self.max_rows = max_rows(height, tile_rows, world)
        self.local = torch.zeros((self.max_rows, width, 3), dtype=torch.float64, device=device)
        self.image = None
        self.parts = None
        self.gathered = None
        self.source_row = None
        if rank == dst:
            self.image = torch.zeros((height, width, 3), dtype=torch.float64, device=device)
            # ONE buffer [rank][max_rows][W][3] (the gather's output list is views of it) and, for every global row,
            # where it sits in that buffer: rank (i // tile_rows) % world, local row (i // tile_rows // world) * tile_rows
            # + i % tile_rows — so the strips go back to their rows with ONE index_select (round 4; it was one
            # index_copy_ launch per rank)
            self.gathered = torch.empty((world, self.max_rows, width, 3), dtype=torch.float64, device=device)
            self.parts = [self.gathered[r] for r in range(world)]
            src = [((i // tile_rows) % world) * self.max_rows + (i // tile_rows // world) * tile_rows + i % tile_rows
                   for i in range(height)]
            self.source_row = torch.tensor(src, dtype=torch.long, device=device)

    def gather(self):
        """local strips -> full image on rank dst (returns it there, None elsewhere)."""
        if self.world == 1:
            self.image[torch.tensor(self.rows, dtype=torch.long, device=self.local.device)] = \
                self.local[: len(self.rows)]
            return self.image
        dist.gather(self.local, self.parts if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        torch.index_select(self.gathered.view(self.world * self.max_rows, self.w, 3), 0, self.source_row, out=self.image)
        return self.image


# ---- sample-split decomposition ------------------------------------
# The reference's own parallelism: every worker renders the WHOLE frame with spp/N of the
# samples and the partial frames are added in worker order (src/render.cpp:169-180).  Here a
# worker is a rank and its samples are a contiguous range of streams.  Unlike the strip
# partition, the result depends on N in the last bits (the sum is re-associated), exactly as
# the reference's result depends on its thread count; and the message is N full frames.


def stream_range(nstreams: int, world: int, rank: int):
    """(first, count) of the streams rank `rank` renders: contiguous, sizes differ by at most 1."""
    base, extra = divmod(nstreams, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


class FrameSum:
    """Gather full frames from all ranks and add them in rank order on rank dst."""

    def __init__(self, height, width, rank, world, device, dst=0):
        self.rank, self.world, self.dst = rank, world, dst
        self.local = torch.zeros((height, width, 3), dtype=torch.float64, device=device)
        self.parts = [torch.empty_like(self.local) for _ in range(world)] if rank == dst else None
        self.image = torch.zeros_like(self.local) if rank == dst else None

    def reduce(self):
        if self.world == 1:
            self.image.copy_(self.local)
            return self.image
        dist.gather(self.local, self.parts if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        self.image.zero_()
        for r in range(self.world):  # global = done + global, workers in launch order (render.cpp:176-180)
            torch.add(self.parts[r], self.image, out=self.image)
        return self.image
