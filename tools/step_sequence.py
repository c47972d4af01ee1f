#!/usr/bin/env python3
"""Wall time of each step (reset + render of the configuration's samples) after upload, one by one: how many
launches k_batch's feedback needs to settle.    python tools/step_sequence.py c3 [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = scenes.CONFIGS[name]()
with abi.Context(cfg.width, cfg.height, 0) as ctx:
    ctx.upload_scene(cfg.scene)
    ctx.set_camera(cfg.camera)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        ctx.reset()
        ctx.render(cfg.spp, 0)
        ts.append((time.perf_counter() - t0) * 1e3)
    st = ctx.stats()
    print(name, "RT_GRAB=" + os.environ.get("RT_GRAB", "default"), " ".join(f"{t:.2f}" for t in ts), "ms; listed", st.batch_listed, "early", st.batch_early, flush=True)
