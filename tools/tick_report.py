#!/usr/bin/env python3
"""How long k_batch's waves wait for dispenser tickets (diagnostic build: bash tools/build_variant.sh tick -DRT_TICKWAIT):

    RT_AMD_LIB=build/tick/librt_amd.so python tools/tick_report.py c3 [spp]

whole frame and one rank of eight; shares of the summed wave lifetime."""
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
    L.rt_debug_tick.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    with abi.Context(cfg.width, cfg.height, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        for world in (1, 8):
            ctx.set_partition(world - 1, world, 64)
            for _ in range(3):
                ctx.reset(); ctx.render(spp, 0)
            L.rt_debug_tick(None, 1)
            ctx.reset(); ctx.render(spp, 0)
            out = (C.c_uint64 * 48)()
            L.rt_debug_tick(out, 0)
            life, w1, w2, n1, n2, ta, tb, tc, nu = [int(out[i]) for i in range(9)]
            print(f"{name} {spp} spp world {world}: wave lifetime {life} ticks over all waves")
            print(f"   tickets drawn a unit ahead: {n1:9d}  waited {100.0 * w1 / life:6.2f} % of lifetime, {w1 / max(n1, 1):8.1f} ticks each")
            print(f"   tickets drawn on the spot:  {n2:9d}  waited {100.0 * w2 / life:6.2f} % of lifetime, {w2 / max(n2, 1):8.1f} ticks each")
            print(f"   units {nu}: phases P + H {ta / max(nu, 1):9.0f} ticks each ({100.0 * ta / life:5.1f} %), phase S {tb / max(nu, 1):9.0f} ({100.0 * tb / life:5.1f} %), "
                  f"epilogue + feedback {tc / max(nu, 1):9.0f} ({100.0 * tc / life:5.1f} %), outside units {100.0 * (life - ta - tb - tc) / life:5.1f} %", flush=True)
            names = ["listed pixel", "early unit", "own band", "helped band", "late units", "own band, first 4", "helped band, first 4", "late, first 4"]
            for k, nm in enumerate(names):
                a, b, c, n = [int(out[16 + 4 * k + i]) for i in range(4)]
                if n:
                    print(f"   {nm:22s} {n:8d} units ({100.0 * (a + b + c) / life:5.1f} % of lifetime): P + H {a / n:8.0f}  S {b / n:8.0f}  E {c / n:8.0f}  all {(a + b + c) / n:8.0f}", flush=True)


if __name__ == "__main__":
    main()
