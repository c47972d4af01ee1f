#!/usr/bin/env python3
"""Where a five-wave build's frame differs from the four-wave build's (diagnostic):
    RT_AMD_LIB=<lib> python tools/five_wave_diff.py [spp] [launches]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402


def render(cfg, waves, spp, launches):
    os.environ["RT_BATCH_WAVES"] = waves
    with abi.Context(cfg.width, cfg.height, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        for _ in range(launches):
            ctx.render(spp, 0)
        st = ctx.stats()
        return ctx.read_framebuffer(), st


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cfg = scenes.c3(64, 48, lon=48, lat=40, ground=(10, 6))
    a, sa = render(cfg, "4", spp, launches)
    b, sb = render(cfg, "5", spp, launches)
    d_rgba = np.any(a["rgba"].view(np.uint32) != b["rgba"].view(np.uint32), axis=-1)
    d_id = a["prim_id"] != b["prim_id"]
    d_t = a["t"].view(np.uint32) != b["t"].view(np.uint32)
    hit = a["prim_id"] >= 0
    print(f"spp {spp} x {launches}: pixels {d_rgba.size}, rgba differs {int(d_rgba.sum())} (of them on hits {int((d_rgba & hit).sum())}), "
          f"prim_id differs {int(d_id.sum())}, t differs {int(d_t.sum())}; shadow rays {sa.shadow_rays} vs {sb.shadow_rays}")
    ys, xs = np.nonzero(d_rgba)
    for y, x in list(zip(ys, xs))[:12]:
        print(f"   ({x},{y}) prim {a['prim_id'][y, x]} / {b['prim_id'][y, x]}  rgba {a['rgba'][y, x][:3]} / {b['rgba'][y, x][:3]}")
    rows = np.nonzero(d_rgba.any(axis=1))[0]
    print("   rows with differences:", rows[:40], "...")


if __name__ == "__main__":
    main()
