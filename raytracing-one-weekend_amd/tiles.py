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
        self.index = None
        if rank == dst:
            self.image = torch.zeros((height, width, 3), dtype=torch.float64, device=device)
            self.parts = [torch.empty_like(self.local) for _ in range(world)]
            self.index = [torch.tensor(strip_rows(height, tile_rows, world, r), dtype=torch.long, device=device)
                          for r in range(world)]

    def gather(self):
        """local strips -> full image on rank dst (returns it there, None elsewhere)."""
        if self.world == 1:
            self.image[torch.tensor(self.rows, dtype=torch.long, device=self.local.device)] = \
                self.local[: len(self.rows)]
            return self.image
        dist.gather(self.local, self.parts if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        for r in range(self.world):
            n = self.index[r].numel()
            if n:
                self.image.index_copy_(0, self.index[r], self.parts[r][:n])
        return self.image
