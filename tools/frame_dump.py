#!/usr/bin/env python3
"""Dump one launch's prim_id / t / rgba of a configuration (for comparing two builds of the library):
    RT_AMD_LIB=<lib> python tools/frame_dump.py c3 64 gpurun_out/x.npz [launches]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402

name, spp, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cfg = scenes.CONFIGS[name]()
with abi.Context(cfg.width, cfg.height, 0) as ctx:
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    for _ in range(launches):
        ctx.render(spp, 0)
    fb = ctx.read_framebuffer()
    st = ctx.stats()
    print(name, spp, "launches", st.batch_launches, "listed", st.batch_listed, "early", st.batch_early, flush=True)
np.savez_compressed(out, prim_id=fb["prim_id"], t=fb["t"], rgba=fb["rgba"])
