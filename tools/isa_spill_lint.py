#!/usr/bin/env python3
"""A check of the compiler's output (raytracer_amd/csrc/rt_kernels.s, `make asm`): spill code under the wrong lane mask.

The register allocator of this ROCm's compiler can place a VGPR spill store or reload at the top of the block in
which a divergent branch rejoins -- BEFORE the `s_or_b64 exec, exec, s[..]` that gives the lanes back.  The store then
saves only the lanes that took the branch (lane 0 behind an `if (lane == 0)`), and the reload after it hands every
other lane what the slot held before: stale values of an earlier unit.  This is what made builds of k_batch render
wrong frames after source changes that have nothing to do with the arithmetic ("five-wave fragility").

    python tools/isa_spill_lint.py [rt_kernels.s]     exit status 1 if any kernel has such a place

Flagged: any scratch_store / scratch_load marked "Folded Spill" / "Folded Reload" that stands between a label (where
branches rejoin) and an exec-restoring `s_or_b64 exec, exec, ...`, with no branch and no other write of exec in
between.  (Spill code inside a divergent region -- behind the s_and_saveexec that opens it -- is not this.)"""
import re
import sys


def lint(path):
    kernel, block, pending, bad, at_top = None, None, [], [], False
    for no, line in enumerate(open(path), 1):
        t = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, block, pending, at_top = m.group(1), m.group(1), [], False
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            block, pending, at_top = m.group(1), [], True      # a label: where branches rejoin
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if "Folded Spill" in t or "Folded Reload" in t:
            if at_top and (t.startswith("scratch_") or t.startswith("buffer_")):
                pending.append((no, t.split(";")[0].strip()))
            continue
        # (kernels with the whole register file to one wave spill into AGPRs: the copies are lane-masked just the same)
        if at_top and re.match(r"v_accvgpr_(write|read)_b32", t):
            pending.append((no, t.split(";")[0].strip()))
            continue
        if re.match(r"s_or_b64\s+exec,\s*exec,", t):
            for p in pending:
                bad.append((kernel, block, p[0], p[1], no))
            pending, at_top = [], False
            continue
        # anything else that writes exec narrows it: spill code behind it belongs to the region it opens
        if re.match(r"s_(cbranch|branch|endpgm|setpc)", t) or re.match(r"s_\w+\s+exec,", t) or "saveexec" in t:
            pending, at_top = [], False
    return bad


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "raytracer_amd/csrc/rt_kernels.s"
    bad = lint(path)
    for k, b, no, ins, at in bad:
        print(f"{path}:{no}: {ins}   -- before the exec restore at line {at}, block {b} of {k}")
    print(f"{len(bad)} spill instruction(s) under a partial lane mask")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
