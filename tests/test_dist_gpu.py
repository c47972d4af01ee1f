"""The N > 1 path end to end on one GPU: two processes (ranks), each with its own context on
cuda:0, render their tile partitions of one frame, export them with rt_export_local_rgba and
assemble the frame with FrameGather over torch.distributed (gloo here; the bench uses nccl =
RCCL with one GPU per rank).  The assembled frame must equal the single-context frame bit
for bit -- over a shallow tree (k_path) and a deep one (the depth-0 pipeline)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from raytracer_amd import abi, scenes

pytestmark = pytest.mark.gpu

W, H, TILE, SPP = 320, 200, 64, 3


def _config(kind):
    return scenes.c2(W, H) if kind == "shallow" else scenes.c3(W, H, f=12, ground=(10, 6))


def _worker(rank, world, port, kind, out_path):
    import torch.distributed as dist
    from raytracer_amd.dist import FrameGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = _config(kind)
    ctx = abi.Context(W, H, 0)
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    ctx.set_partition(rank, world, TILE)
    ctx.render(SPP, 0)
    g = FrameGather(W, H, world, TILE, "cpu", abi.partition_pixels)
    stage = torch.zeros((g.n_max, 4), dtype=torch.float32, device="cuda")
    ctx.export_local_rgba(stage.data_ptr())                 # device-to-device, as under RCCL
    g.local.copy_(stage)
    frame = g.gather(dist)
    if rank == 0:
        np.save(out_path, frame.numpy())
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["shallow", "deep"])
def test_two_ranks_render_export_gather(tmp_path, kind):
    cfg = _config(kind)
    with abi.Context(W, H, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        ctx.render(SPP, 0)
        ref = ctx.read_framebuffer()["rgba"]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, port, kind, out), nprocs=2, join=True)
    frame = np.load(out)
    assert frame.shape == ref.shape
    assert np.array_equal(frame.view(np.uint32), ref.view(np.uint32))
    assert (ref[..., :3] > 0).any()


def test_bench_collective_branch_over_rccl_with_one_rank():
    """bench.py's N > 1 branch (nccl = RCCL process group, export into the collective's device buffer,
    all_gather_into_tensor, barrier, all-reduces) run on the one GPU at hand with a world of one: the
    frame the all-gather assembled equals the context's framebuffer bit for bit."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "c3", "--width", "320",
                        "--height", "200", "--spp", "3", "--steps", "2", "--warmup", "1", "--no-cpu",
                        "--collective-with-one-rank"], cwd=root, env=env, capture_output=True, text=True,
                       timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0
    reh = out["collective_rehearsal"]
    assert reh["backend"] == "nccl" and reh["frame_equals_framebuffer"] is True
