#!/usr/bin/env python3
"""When the waves of a k_batch launch end (diagnostic build, tools/build_stamps.sh):

    RT_AMD_LIB=build/stamps/librt_amd.so python tools/wave_timeline.py c3 [world] [rank] [spp]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cfg = scenes.CONFIGS[name]()
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else min(cfg.spp, 64)
    L = abi.lib()
    L.rt_debug_wave_log.argtypes = [C.c_void_p]
    with abi.Context(cfg.width, cfg.height, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        ctx.set_partition(rank, world, 64)
        for _ in range(3):
            ctx.reset()
            ctx.render(spp, 0)
        log = np.zeros((8, 32768), dtype=np.uint64)
        L.rt_debug_wave_log(log.ctypes.data)
    used = log[1] > 0
    b, e, units, meta = log[0][used].astype(np.int64), log[1][used].astype(np.int64), log[2][used], log[3][used]
    t0 = b.min()
    b, e = (b - t0) / 100.0, (e - t0) / 100.0          # microseconds
    print(f"{name} world {world} rank {rank}: {used.sum()} waves, first start 0, last start {b.max():.0f} us, "
          f"last end {e.max():.0f} us, mean end {e.mean():.0f} us, sum of lifetimes / (waves x last end) = {(e - b).sum() / (used.sum() * e.max()):.3f}")
    edges = np.linspace(0, e.max(), 21)
    hist, _ = np.histogram(e, bins=edges)
    for i, h in enumerate(hist):
        print(f"  waves ending in {edges[i]:7.0f}..{edges[i + 1]:7.0f} us: {h:5d}")
    xcd = (meta & 15).astype(int)
    for x in range(8):
        m = xcd == x
        if m.any():
            print(f"  XCD {x}: {m.sum():5d} waves, units {int(units[m].sum()):7d}, listed tasks {int((meta[m] >> 16).sum()):5d}, "
                  f"last end {e[m].max():7.0f} us, mean end {e[m].mean():7.0f} us")
    ls = (log[4][used].astype(np.int64) - t0) / 100.0
    lu, lc = log[5][used], log[6][used]
    late = np.argsort(-e)[:16]
    for i in late:
        print(f"  late wave: end {e[i]:.0f} us, units {int(units[i])}, xcd {int(meta[i] & 15)}, tasks {int(meta[i] >> 16)}; its last unit: "
              f"{int(lu[i] & 0xffffffff)} (stage {int(lu[i] >> 32) - 2}) started {ls[i]:.0f} us, ran {e[i] - ls[i]:.0f} us, cost {int(lc[i])}")
    d = e - ls
    print(f"  last units of all waves: mean duration {d.mean():.0f} us, mean cost {lc.mean():.0f}; started after the first wave ended: {(ls > e.min()).sum()}")


if __name__ == "__main__":
    main()
