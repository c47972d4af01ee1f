#!/usr/bin/env python3
"""Repeat a small deep-tree render with fresh contexts and compare against the first: hunts ordering races."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes

def run(cfg, launches, spp):
    with abi.Context(cfg.width, cfg.height, 0) as ctx:
        ctx.upload_scene(cfg.scene); ctx.set_camera(cfg.camera)
        for _ in range(launches):
            ctx.render(spp, 0)
        st = ctx.stats()
        return ctx.read_framebuffer(), st

def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    for (w, h) in ((64, 48), (160, 96), (320, 200)):
        cfg = scenes.c3(w, h, lon=48, lat=40, ground=(10, 6))
        ref = None
        bad = 0
        for i in range(reps):
            got, st = run(cfg, 3, 64)
            if ref is None:
                ref = got
            else:
                d = (got["prim_id"] != ref["prim_id"]).sum() + (got["t"].view(np.uint32) != ref["t"].view(np.uint32)).sum() + (got["rgba"].view(np.uint32) != ref["rgba"].view(np.uint32)).sum()
                bad += int(d > 0)
                if d:
                    print(f"  {w}x{h} rep {i}: {int(d)} differing values, shadow rays {st.shadow_rays}")
        print(f"{w}x{h}: {bad} of {reps - 1} repeats differ from the first; env " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("RT_")), flush=True)

if __name__ == "__main__":
    main()
