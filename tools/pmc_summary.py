#!/usr/bin/env python3
"""Sum every counter of the given rocprofv3 counter_collection.csv files per kernel (short
name) and print one line per (kernel, counter): total and per launch.

  python tools/pmc_summary.py <kernel substring> file.csv [file.csv ...]"""
import collections
import csv
import sys


def main():
    want = sys.argv[1]
    for path in sys.argv[2:]:
        tot = collections.defaultdict(float)
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(path)):
            if want not in r["Kernel_Name"]:
                continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Counter_Name"]].add(r["Dispatch_Id"])
        print(path)
        for k in sorted(tot):
            n = max(len(disp[k]), 1)
            print("  %-28s launches %4d  total %16.0f  per launch %14.0f" % (k, n, tot[k], tot[k] / n))


if __name__ == "__main__":
    main()
