"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on
ROCm, "gloo" for the CPU rehearsal).  Pixels shard naturally -- the frame is split into
64x64 tiles dealt round-robin to the ranks, the scene is replicated, and the ONLY
collective is one gather of the partition framebuffers after the last sample
(SURVEY.md section 8(e)).  Nothing here computes radiance."""
from __future__ import annotations

import numpy as np


def env_rank_world():
    import os
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


class FrameGather:
    """All-gathers per-rank [n_local, 4] RGBA blocks (padded to the largest partition) and
    scatters them into the full [H, W, 4] frame by each rank's pixel indices."""

    def __init__(self, width, height, world, tile, device, partition_fn):
        import torch
        self.torch = torch
        self.width, self.height, self.world = width, height, world
        parts = [np.asarray(partition_fn(width, height, r, world, tile)) for r in range(world)]
        self.counts = [int(p.shape[0]) for p in parts]
        self.n_max = max(self.counts) if parts else 0
        idx = np.full((world, self.n_max), width * height, np.int64)   # pad -> dump slot
        for r, p in enumerate(parts):
            idx[r, :p.shape[0]] = p
        self.index = torch.from_numpy(idx.reshape(-1)).to(device)
        self.local = torch.zeros((self.n_max, 4), dtype=torch.float32, device=device)
        self.gathered = torch.zeros((world * self.n_max, 4), dtype=torch.float32, device=device)
        self.frame = torch.zeros((width * height + 1, 4), dtype=torch.float32, device=device)

    def gather(self, dist=None):
        """`self.local[:n_local]` must hold this rank's RGBA; returns the full frame tensor
        [H, W, 4] (valid on every rank: an all-gather keeps the ranks symmetric)."""
        if dist is not None:             # also with a world of one: the collective itself runs
            dist.all_gather_into_tensor(self.gathered, self.local)
        else:
            self.gathered.copy_(self.local)
        self.frame.index_copy_(0, self.index, self.gathered)
        return self.frame[:-1].view(self.height, self.width, 4)
