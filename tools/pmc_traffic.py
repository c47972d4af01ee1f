#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into HBM bytes per launch per kernel.

  python tools/pmc_traffic.py <config> <fetch_counter_collection.csv> <write_counter_collection.csv> \
         [out.json]

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes
(guide, "HBM"): it is doubled here.  Access widths other than 16 B/lane are uncalibrated in the
guide; the kernels here read 4 B/lane, so treat the absolute as approximate."""
import collections
import csv
import json
import os
import sys


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        short = k.split("(")[0].split("::")[-1].split("<")[0].replace("void ", "").strip()
        tot[short] += float(r["Counter_Value"])
        disp[short].add(r["Dispatch_Id"])
    return {k: (tot[k], len(disp[k])) for k in tot}


def main():
    cfg, fpath, wpath = sys.argv[1:4]
    out = sys.argv[4] if len(sys.argv) > 4 else None
    f = per_kernel(fpath, "FETCH_SIZE")
    w = per_kernel(wpath, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fk, fn = f.get(k, (0.0, 0))
        wk, wn = w.get(k, (0.0, 0))
        n = max(fn, wn, 1)
        rd = 2.0 * fk * 1024.0 / n
        wr = wk * 1024.0 / n
        res[k] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                  "bytes_per_launch": rd + wr}
    if out:
        d = {}
        if os.path.exists(out):
            d = json.load(open(out))
        d[cfg] = res
        json.dump(d, open(out, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        print("%-22s launches %4d  read %12.0f B  write %12.0f B per launch" %
              (k, v["launches"], v["read_bytes_per_launch"], v["write_bytes_per_launch"]))


if __name__ == "__main__":
    main()
