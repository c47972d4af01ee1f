#!/usr/bin/env python3
"""Basic blocks of one kernel in a hipcc -save-temps / -S listing with their instruction classes:
    python tools/isa_blocks.py raytracer_amd/csrc/rt_kernels.s k_batchILb1ELi4 [min_insts]
Per block: label, line, VALU / SALU / SMEM / LDS / VMEM counts, v_readlane + v_writelane (SGPR spill traffic) and
scratch_ loads / stores (VGPR spills), and where it branches -- enough to see which spills sit in a hot loop."""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    mn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    on, blocks, cur = False, [], None
    for n, line in enumerate(open(path), 1):
        s = line.strip()
        if not on:
            if re.match(r"^_Z\w*%s\w*:" % key, s):
                on = True
                cur = {"label": "entry", "line": n, "c": {}, "br": []}
                blocks.append(cur)
            continue
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = {"label": m.group(1), "line": n, "c": {}, "br": []}
            blocks.append(cur)
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        c = cur["c"]
        def add(k): c[k] = c.get(k, 0) + 1
        if op.startswith("v_readlane") or op.startswith("v_writelane"):
            add("lane")
        if op.startswith("scratch_"):
            add("scr")
        elif op.startswith("v_"):
            add("valu")
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            add("smem")
        elif op.startswith("s_cbranch") or op.startswith("s_branch"):
            cur["br"].append(s.split()[-1])
            add("salu")
        elif op.startswith("s_"):
            add("salu")
        elif op.startswith("ds_"):
            add("lds")
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
            add("vmem")
    for b in blocks:
        c = b["c"]
        tot = sum(v for k, v in c.items() if k != "lane")
        if tot < mn:
            continue
        print("%-12s @%-6d valu %4d salu %4d smem %3d lds %3d vmem %3d | lane %3d scratch %3d -> %s"
              % (b["label"], b["line"], c.get("valu", 0), c.get("salu", 0), c.get("smem", 0), c.get("lds", 0),
                 c.get("vmem", 0), c.get("lane", 0), c.get("scr", 0), " ".join(b["br"])))


if __name__ == "__main__":
    main()
