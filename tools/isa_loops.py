#!/usr/bin/env python3
"""Instruction classes per loop of one kernel in the `make asm` listing:
    python tools/isa_loops.py raytracer_amd/csrc/rt_kernels.s k_batchILb1ELi5 [min_depth]
`lane` = v_readlane + v_writelane (SGPR spill traffic when it sits in an inner loop), `scr` = scratch accesses."""
import re
import sys


def agg(path, key):
    on, depth, hdr, tot = False, 0, None, {}
    for line in open(path):
        s = line.strip()
        if not on:
            on = bool(re.match(r"^_Z\w*%s\w*:" % key, s))
            continue
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", s)
        if m:
            d = re.search(r"Depth=(\d+)", m.group(2))
            h = re.search(r"Header=(BB\d+_\d+)", m.group(2))
            depth = int(d.group(1)) if d else 0
            hdr = h.group(1) if h else (m.group(1)[2:] if d else None)
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        t = tot.setdefault((depth, hdr), {})
        def add(n): t[n] = t.get(n, 0) + 1
        if op.startswith("v_readlane") or op.startswith("v_writelane"):
            add("lane")
        if op.startswith("scratch_"):
            add("scr")
        elif op.startswith("v_"):
            add("valu")
        elif op.startswith("s_load"):
            add("smem")
        elif op.startswith("s_"):
            add("salu")
        elif op.startswith("ds_"):
            add("lds")
        else:
            add("vmem")
    return tot


if __name__ == "__main__":
    t = agg(sys.argv[1], sys.argv[2])
    mn = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    for k in sorted(t, key=lambda k: (k[0], str(k[1]))):
        if k[0] >= mn:
            print("  depth", k[0], k[1], " ".join("%s %d" % (n, t[k][n]) for n in sorted(t[k])))
