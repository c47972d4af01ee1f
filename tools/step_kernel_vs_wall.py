#!/usr/bin/env python3
"""Wall time of a step (reset + render) against the k_batch kernel time inside it (HIP events on the lane's stream),
whole frame and one rank of eight.     python tools/step_kernel_vs_wall.py c3 [spp] [steps]"""
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
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    with abi.Context(cfg.width, cfg.height, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        for world in (1, 8):
            for rank in sorted({0, world - 1}):
                ctx.set_partition(rank, world, 64)
                for _ in range(3):
                    ctx.reset(); ctx.render(spp, 0)
                for timing in (False, True):
                    ctx.reset_stats()
                    ctx.set_timing(timing)
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        ctx.reset(); ctx.render(spp, 0)
                    wall = (time.perf_counter() - t0) / steps * 1e3
                    st = ctx.stats()
                    k = st.path_ms / max(1, st.path_launches) if timing else float("nan")
                    print(f"{name} {spp} spp world {world} rank {rank} timing {int(timing)}: wall {wall:7.3f} ms/step  k_batch {k:7.3f} ms/launch  "
                          f"({st.path_launches // steps if timing else '-'} launches/step)", flush=True)
                ctx.set_timing(False)


if __name__ == "__main__":
    main()
