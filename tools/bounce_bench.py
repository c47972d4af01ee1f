#!/usr/bin/env python3
"""Throughput of the path kernel WITH bounces (k_path<.., DEPTH0 = false>) on the bench scenes at 1080p:
not a bench configuration (the metric is quoted at max_depth 0); numbers for DESIGN 4.2.

    python tools/bounce_bench.py [depth] [spp] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402


def main():
    depth = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    for name in os.environ.get("BOUNCE_CONFIGS", "c2,c3").split(","):
        cfg = scenes.CONFIGS[name](1920, 1080)
        with abi.Context(cfg.width, cfg.height, 0) as ctx:
            ctx.upload_scene(cfg.scene)
            ctx.set_camera(cfg.camera)
            ctx.render(4, depth)                     # warm-up: feedback sets, caches
            ctx.reset()
            ctx.reset_stats()
            t0 = time.perf_counter()
            for _ in range(steps):
                ctx.render(spp, depth)
            dt = time.perf_counter() - t0
            st = ctx.stats()
            rays = st.primary_rays + st.shadow_rays + st.bounce_rays
            print(f"{name} depth {depth}: {dt / steps / spp * 1e3:7.3f} ms per sample  {rays / dt / 1e6:9.1f} Mrays/s  "
                  f"({rays / (steps * spp * cfg.width * cfg.height):.2f} rays per pixel-sample, coop chunks {st.coop_chunks})", flush=True)


if __name__ == "__main__":
    main()
