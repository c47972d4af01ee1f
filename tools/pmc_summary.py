#!/usr/bin/env python3
"""Sum every counter of the given rocprofv3 counter_collection.csv files for the kernels whose
short name (namespace, template arguments and parameters stripped) is one of the given names;
prints a "== kernel" section per name with one line per counter: total and per launch.

  python tools/pmc_summary.py <kernel[,kernel...]> file.csv [file.csv ...]"""
import collections
import csv
import sys


def short_name(k):
    return k.split("(")[0].split("<")[0].split("::")[-1].replace("void ", "").strip()


def main():
    for want in sys.argv[1].split(","):
        tot = collections.defaultdict(float)
        disp = collections.defaultdict(set)
        source = {}                      # a counter collected in several passes is taken from the first
        for path in sys.argv[2:]:
            for r in csv.DictReader(open(path)):
                if want != short_name(r["Kernel_Name"]):
                    continue
                if source.setdefault(r["Counter_Name"], path) != path:
                    continue
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                disp[r["Counter_Name"]].add(r["Dispatch_Id"])
        if not tot:
            continue
        print("== %s" % want)
        for k in sorted(tot):
            n = max(len(disp[k]), 1)
            print("  %-28s launches %4d  total %16.0f  per launch %14.0f" % (k, n, tot[k], tot[k] / n))


if __name__ == "__main__":
    main()
