#!/usr/bin/env python3
"""Where a wave's cycles go in the batch kernels (diagnostic build, tools/build_stamps.sh):

    RT_AMD_LIB=build/stamps/librt_amd.so python tools/stamps_report.py c3 [spp]

renders one warm-up and one measured step of the config at max_depth 0 and prints, per kernel,
the share of the summed wave lifetime spent in each stamped segment."""
import ctypes as C
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402

SEG = ["wave lifetime", "dispenser", "-", "-",
       "stash save/load", "(traverse)", "node-group load wait", "node tests + ordering",
       "triangle load wait", "triangle tests", "unit tail (feedback)", "units", "traversals"]
SEG1 = ["prologue loads until arrived", "ray generation (Philox, camera ray, pair set-up)",
        "hit record + BRDF-sample verdict", "prediction check / chain walk, records to LDS",
        "phase S: record loads", "light sampling + shadow pair set-up", "blocked verdict, radiance",
        "epilogue: shares to LDS", "epilogue: running-mean chain + texel store"]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    cfg = scenes.CONFIGS[name]()
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else min(cfg.spp, 64)
    L = abi.lib()
    L.rt_debug_stamps.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    world = int(os.environ.get("STAMPS_WORLD", "1"))
    with abi.Context(cfg.width, cfg.height, 0) as ctx:
        ctx.upload_scene(cfg.scene)
        ctx.set_camera(cfg.camera)
        ctx.set_partition(int(os.environ.get("STAMPS_RANK", "0")), world, 64)
        for _ in range(2):                  # two warm-ups: a launch lists costly units with the threshold it takes from
            ctx.render(spp, 0)              # the launch before it, so the third launch is the first to hand a list out
            ctx.reset()
        L.rt_debug_stamps(None, 1)
        ctx.render(spp, 0)
        out = (C.c_uint64 * 64)()
        L.rt_debug_stamps(out, 0)
    hist = [int(out[48 + i]) for i in range(16)]
    cyc = [int(out[i]) for i in range(16)]
    if sum(hist):
        print("unit cost (node groups + triangle records) -> units, share of all unit cycles, mean cycles")
        for b in range(16):
            if hist[b]:
                lo = 0 if b == 0 else 1 << (b - 1)
                print(f"  {lo:6d}..{(1 << b) - 1:6d}: {hist[b]:9d} units  {100.0 * cyc[b] / max(sum(cyc), 1):6.2f} % of cycles  "
                      f"{cyc[b] / hist[b]:10.0f} cycles each")
    for k, kname in enumerate(["-", "-", "k_batch", "-"]):
        a = [int(out[16 * k + i]) for i in range(16)]
        if not a[0] or k != 2:
            continue
        print(f"== {kname} ({name}, {spp} spp): {a[11]} units, {a[12]} traversals, "
              f"{a[0] / max(a[11], 1):.0f} cycles of wave lifetime per unit")
        b = [int(out[16 + i]) for i in range(16)]
        for i in range(1, 11):
            if a[i]:
                print(f"  {SEG[i]:48s} {100.0 * a[i] / a[0]:6.2f} %   {a[i] / max(a[11], 1):10.0f} cycles / unit")
        for i in range(len(SEG1)):
            if b[i] and i < 10:
                print(f"  {SEG1[i]:48s} {100.0 * b[i] / a[0]:6.2f} %   {b[i] / max(a[11], 1):10.0f} cycles / unit")
        print(f"  {'unaccounted':48s} {100.0 * (a[0] - sum(a[1:11]) - sum(b[:10])) / a[0]:6.2f} %")
        cyc, cost, unit = a[13] >> 32, (a[13] >> 20) & 0xfff, a[13] & 0xfffff
        t0 = (~b[12]) & 0xffffffffffffffff
        if b[12]:
            print(f"  launch timeline (units of 10 ns from the first wave's start): last listed pixel task ends at {b[13] - t0 if b[13] else 0}, "
                  f"last wave at {b[14] - t0}; {b[15]} listed pixel tasks, mean {b[11] / max(b[15], 1):.0f} cycles, longest {b[10]}")
        print(f"  longest single unit: {cyc} cycles ({cyc / 2.4e3:.0f} us at 2.4 GHz), unit {unit} of its launch, "
              f"{cost} node groups + triangle records; open pixels enumerated: {a[14]}, enumeration passes: {a[15]}")


if __name__ == "__main__":
    main()
