#!/usr/bin/env python3
"""Where the part of a step that does not scale with the sample count goes: reset alone, then reset + render(n, 0)
for n = 1 .. 64, whole frame and one rank of eight (host wall clock around blocking calls, best of `reps`).

    python tools/step_fixed_cost.py c3 [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402


def best(f, reps):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        b = min(b, time.perf_counter() - t0)
    return b * 1e3


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cfg = scenes.CONFIGS[name]()
    with abi.Context(cfg.width, cfg.height, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        for world in (1, 8):
            ctx.set_partition(world - 1, world, 64)
            for _ in range(3):
                ctx.reset(); ctx.render(64, 0)
            t_reset = best(lambda: ctx.reset(), reps)
            print(f"{name} world {world}: reset alone {t_reset:.3f} ms", flush=True)
            for n in (1, 2, 4, 8, 16, 32, 64):
                def step():
                    ctx.reset(); ctx.render(n, 0)
                for _ in range(3):
                    step()                               # the feedback lists of this shape settle
                t = best(step, reps)
                def again():
                    ctx.render(n, 0)
                ctx.reset(); ctx.render(n, 0)
                t2 = best(again, reps)                   # further samples without a reset: the pixels' history is warm
                print(f"   render({n:2d}, 0) after reset {t:8.3f} ms   without reset (samples {n}..) {t2:8.3f} ms", flush=True)


if __name__ == "__main__":
    main()
