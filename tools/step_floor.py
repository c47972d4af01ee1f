#!/usr/bin/env python3
"""The floor of a k_batch step: render(n, 0) of ever smaller partitions (one rank of `world`), best of `reps`.

    python tools/step_floor.py c3 [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cfg = scenes.CONFIGS[name]()
    with abi.Context(cfg.width, cfg.height, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        for world in (8, 32, 128, 480):
            ctx.set_partition(world // 2, world, 64)
            for n in (8, 64):
                def step():
                    ctx.reset(); ctx.render(n, 0)
                for _ in range(3):
                    step()
                b = 1e9
                for _ in range(reps):
                    t0 = time.perf_counter(); step(); b = min(b, time.perf_counter() - t0)
                st = ctx.stats()
                print(f"{name} world {world:4d} render({n:2d}, 0): {b * 1e3:8.3f} ms   listed {st.batch_listed} early {st.batch_early}", flush=True)


if __name__ == "__main__":
    main()
