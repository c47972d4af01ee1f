#!/usr/bin/env python3
"""Traversal counters of one step (diagnostic build: bash tools/build_variant.sh stats -DRT_TRAV_STATS):

    RT_AMD_LIB=build/stats/librt_amd.so python tools/trav_stats.py c3 [spp]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    cfg = scenes.CONFIGS[name]()
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else min(cfg.spp, 64)
    L = abi.lib()
    L.rt_debug_trav_stats.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    with abi.Context(cfg.width, cfg.height, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        L.rt_debug_trav_stats(None, 1)
        ctx.reset_stats()
        ctx.render(spp, 0)
        out = (C.c_uint64 * 8)()
        L.rt_debug_trav_stats(out, 0)
        st = ctx.stats()
    t, g, tri, past = (int(out[i]) for i in range(4))
    rays = st.primary_rays + st.shadow_rays
    print(f"{name} {spp} spp: {t} wave traversals, {g} node groups expanded ({g / max(t, 1):.2f} per traversal), "
          f"{tri} triangle records tested ({tri / max(t, 1):.2f} per traversal), {past} went past the window pre-test; "
          f"{rays} rays = {rays / max(t, 1):.1f} per wave traversal of 128 slots")


if __name__ == "__main__":
    main()
