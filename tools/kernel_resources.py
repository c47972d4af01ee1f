#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage and code size of raytracer_amd/csrc/rt_kernels.hip:
runs `make asm` (hipcc -S -Rpass-analysis=kernel-resource-usage) and prints one line per kernel.

    python tools/kernel_resources.py [EXTRA=-D...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout
    return [l.split("(")[0].replace("void ", "").replace("rt::", "") for l in out.splitlines()]


def main():
    r = subprocess.run(["make", "-C", CSRC, "asm"] + sys.argv[1:], capture_output=True, text=True)
    text = r.stderr + r.stdout
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?:\[[^\]]*\])?):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    names = demangle([r["name"] for r in rows])
    print("%-44s %5s %5s %7s %7s %6s %4s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "vspill", "occ"))
    for r, n in zip(rows, names):
        if r.get("VGPRs") is None:
            continue
        print("%-44s %5s %5s %7s %7s %6s %4s" % (n[:44], r.get("VGPRs"), r.get("TotalSGPRs"),
                                                 r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"),
                                                 r.get("VGPRs Spill"), r.get("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main()
