#!/usr/bin/env python3
"""Throughput of the general path kernel (triangle / environment lights, every BRDF, mesh textures)
on the feature scenes at 1080p -- these are not bench configurations; the numbers go into DESIGN 4.3.

    python tools/general_bench.py [spp] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402
from raytracer_amd.scene import CameraArgs  # noqa: E402


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    W, H = 1920, 1080
    wide = dict(near=0.5, pos=(0.0, 20.0, 60.0), lookat=(0.0, 0.0, -400.0))
    cases = [("demo (2 spheres, triangle light)", scenes.demo_scene(), CameraArgs.for_image(W, H)),
             ("demo + environment light", scenes.demo_scene_with_environment(), CameraArgs.for_image(W, H)),
             ("area (410 prims, 5 lights of 3 kinds)", scenes.area_test_scene()[0], CameraArgs.for_image(W, H, **wide)),
             ("materials (every BRDF)", scenes.material_test_scene()[0], CameraArgs.for_image(W, H, **wide)),
             ("textured (every BRDF, 2 textures)", scenes.textured_test_scene()[0], CameraArgs.for_image(W, H, **wide))]
    only = os.environ.get("GENERAL_ONLY")
    with abi.Context(W, H, 0) as ctx:
        for name, scene, cam in cases:
            if only and only not in name:
                continue
            ctx.upload_scene(scene)
            ctx.set_camera(cam)
            for depth in (0, 6):
                ctx.reset()
                ctx.render(1, depth)
                ctx.reset()
                ctx.reset_stats()
                t0 = time.perf_counter()
                for _ in range(steps):
                    ctx.render(spp, depth)
                dt = time.perf_counter() - t0
                st = ctx.stats()
                rays = st.primary_rays + st.shadow_rays + st.bounce_rays
                print(f"{name:40s} depth {depth}: {dt / steps / spp * 1e3:7.3f} ms per sample  "
                      f"{rays / dt / 1e6:9.1f} Mrays/s  ({rays / (steps * spp * W * H):.2f} rays per pixel-sample)", flush=True)


if __name__ == "__main__":
    main()
