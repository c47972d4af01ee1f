#!/usr/bin/env python3
"""Per-rank step time of a world-way split, measured on ONE GPU (compute only):

    python tools/rank_times.py c3 [world] [tile] [spp] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    tile = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    cfg = scenes.CONFIGS[name]()
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else min(cfg.spp, 64)
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    ts = []
    with abi.Context(cfg.width, cfg.height, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        for rank in range(world):
            ctx.set_partition(rank, world, tile)
            for _ in range(3):
                ctx.reset()
                ctx.render(spp, 0)
            ctx.reset_stats()
            t0 = time.perf_counter()
            for _ in range(steps):
                ctx.reset()
                ctx.render(spp, 0)
            ts.append((time.perf_counter() - t0) / steps * 1e3)
            st = ctx.stats()
            print(f"{name} world {world} tile {tile} rank {rank}: {ts[-1]:7.3f} ms per step, {ctx.local_pixel_count()} pixels, "
                  f"{(st.primary_rays + st.shadow_rays) / steps / 1e6:.1f} Mrays per step", flush=True)
    print(f"{name} world {world} tile {tile}: worst {max(ts):.3f} ms, mean {sum(ts) / len(ts):.3f} ms, sum {sum(ts):.3f} ms")


if __name__ == "__main__":
    main()
