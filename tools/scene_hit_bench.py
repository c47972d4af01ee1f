#!/usr/bin/env python3
"""rt_scene_hit (Scene::hit on caller rays) on C3: camera-like rays (coherent) and random rays (incoherent),
host buffers in and out (the copies are part of the call).   python tools/scene_hit_bench.py [n_rays]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    cfg = scenes.CONFIGS[os.environ.get("HIT_CONFIG", "c3")]()
    rng = np.random.default_rng(5)
    ext = float(np.abs(cfg.scene.tri_pos).max())
    with abi.Context(64, 64, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        side = int(np.sqrt(n))
        yy, xx = np.mgrid[0:side, 0:side]
        cam_d = np.stack([(xx / side - 0.5) * 0.8, (yy / side - 0.5) * 0.45, -np.ones_like(xx, float)], axis=2).reshape(-1, 3)
        cases = {"camera-like": (np.zeros_like(cam_d), cam_d),
                 "random": (rng.uniform(-ext, ext, (n, 3)) * 0.6, rng.normal(size=(n, 3)))}
        for name, (o, d) in cases.items():
            o = np.ascontiguousarray(o, np.float32)
            d = np.ascontiguousarray(d, np.float32)
            ctx.scene_hit(o, d)
            t0 = time.perf_counter()
            for _ in range(3):
                t, pid, hit = ctx.scene_hit(o, d)
            dt = (time.perf_counter() - t0) / 3
            print(f"{name:12s} {o.shape[0]} rays: {dt * 1e3:8.2f} ms per call  {o.shape[0] / dt / 1e6:8.1f} Mrays/s  hit {hit.mean():.2f}", flush=True)


if __name__ == "__main__":
    main()
