#!/usr/bin/env python3
"""What strong scaling to expect, measured on ONE GPU: render rank 0's partition of a 1-, 2-, 4-
and 8-way split (compute only, no gather) and print ms per step and the speed-up over the whole
frame.  Ranks' partitions are interleaved 64x64 tiles, so rank 0's cost is representative.

    python tools/partition_rehearsal.py c3 [spp] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    cfg = scenes.CONFIGS[name]()
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else min(cfg.spp, 64)
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    worlds = [int(w) for w in os.environ.get("REHEARSAL_WORLDS", "1,2,4,8").split(",")]
    base = None
    with abi.Context(cfg.width, cfg.height, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        for world in worlds:
            worst = 0.0
            for rank in sorted({0, world - 1}):
                ctx.set_partition(rank, world, 64)
                ctx.render(spp, 0)                      # warm-up (allocations, caches)
                ctx.reset()
                t0 = time.perf_counter()
                for _ in range(steps):
                    ctx.reset()
                    ctx.render(spp, 0)
                worst = max(worst, (time.perf_counter() - t0) / steps * 1e3)
            base = base or worst
            print(f"{name} {spp} spp  world {world}: {worst:8.3f} ms per step ({worst / spp:.4f} ms per sample)  "
                  f"speed-up {base / worst:5.2f}x  streams {ctx.stats().streams}", flush=True)


if __name__ == "__main__":
    main()
