"""world_size-2 rehearsal of the multi-GPU gather on CPU (gloo): each rank fills its tile
partition, one all-gather assembles the frame (raytracer_amd/dist.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import abi
from raytracer_amd.dist import FrameGather

W, H, TILE = 200, 136, 64


def _value(idx):
    idx = idx.astype(np.float64)
    return np.stack([idx, idx * 0.5, -idx, np.ones_like(idx)], -1).astype(np.float32)


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = FrameGather(W, H, world, TILE, "cpu", abi.partition_pixels)
    mine = abi.partition_pixels(W, H, rank, world, TILE)
    g.local.zero_()
    g.local[:len(mine)] = torch.from_numpy(_value(mine))
    frame = g.gather(dist)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    frame = np.load(out)
    ref = _value(np.arange(W * H)).reshape(H, W, 4)
    assert np.array_equal(frame, ref)


def test_single_rank_gather():
    g = FrameGather(W, H, 1, TILE, "cpu", abi.partition_pixels)
    mine = abi.partition_pixels(W, H, 0, 1, TILE)
    g.local[:len(mine)] = torch.from_numpy(_value(mine))
    frame = g.gather(None).numpy()
    assert np.array_equal(frame, _value(np.arange(W * H)).reshape(H, W, 4))
