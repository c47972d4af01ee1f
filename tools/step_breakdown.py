#!/usr/bin/env python3
"""Where a step's wall time goes on a partition: host clock around reset + render, HIP events around the
whole rt_render call, HIP events around the kernel launches alone.

    python tools/step_breakdown.py c3 [world] [spp] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cfg = scenes.CONFIGS[name]()
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else min(cfg.spp, 64)
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    with abi.Context(cfg.width, cfg.height, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        ctx.set_partition(0, world, 64)
        for _ in range(2):
            ctx.reset()
            ctx.render(spp, 0)
        ctx.reset_stats()
        ctx.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.reset()
            ctx.render(spp, 0)
        wall = (time.perf_counter() - t0) / steps * 1e3
        st = ctx.stats()
        print(f"{name} world {world} {spp} spp: wall {wall:.3f} ms per step | rt_render events {st.render_ms / steps:.3f} ms | "
              f"kernels {st.path_ms / steps:.3f} ms in {st.path_launches / steps:.1f} launches | listed units {st.batch_listed}")


if __name__ == "__main__":
    main()
