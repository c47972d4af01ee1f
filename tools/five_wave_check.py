#!/usr/bin/env python3
"""Five-wave against four-wave build of k_batch on the small deep scene of test_five_wave_batch_kernel_against_oracle:
frames equal?  stats?   (diagnostic)   python tools/five_wave_check.py [repeats]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    cfg = scenes.c3(64, 48, lon=48, lat=40, ground=(10, 6))
    ref = None
    for rep in range(reps):
        for waves in ("4", "5"):
            os.environ["RT_BATCH_WAVES"] = waves
            with abi.Context(cfg.width, cfg.height, 0) as ctx:
                ctx.upload_scene(cfg.scene)
                ctx.set_camera(cfg.camera)
                ctx.reset_stats()
                for _ in range(3):
                    ctx.render(64, 0)
                st = ctx.stats()
                fb = ctx.read_framebuffer()
            if ref is None:
                ref = fb
            same = all(np.array_equal(ref[k].view(np.uint32), fb[k].view(np.uint32)) for k in ("rgba", "t", "prim_id"))
            print(f"rep {rep} waves {waves}: listed {st.batch_listed} early {st.batch_early} samples {st.samples} "
                  f"primary {st.primary_rays} shadow {st.shadow_rays}  frame equals first: {same}", flush=True)


if __name__ == "__main__":
    main()
