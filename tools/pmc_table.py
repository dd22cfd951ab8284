#!/usr/bin/env python3
"""Median of every collected PMC counter per (kernel, grid) from one or more rocprofv3 --pmc output directories.
    python tools/pmc_table.py <dir> [<dir> ...]      (each dir: a `rocprofv3 --pmc A B C --kernel-trace --output-format csv` run)"""
import collections
import csv
import glob
import os
import statistics
import sys


def main():
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if "gf::" not in r["Kernel_Name"]:
                    continue
                key = (r["Kernel_Name"].replace("void ", "").split("(")[0][:60], int(r.get("Grid_Size", 0) or 0))
                vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for v in vals.values() for c in v})
    print("| kernel | grid | " + " | ".join(counters) + " |")
    print("|---|---|" + "---|" * len(counters))
    for key in sorted(vals, key=lambda k: (k[0], k[1])):
        if max(len(x) for x in vals[key].values()) < 10:
            continue
        row = [f"{statistics.median(vals[key][c]):.4g}" if c in vals[key] else "" for c in counters]
        print(f"| {key[0]} | {key[1]} | " + " | ".join(row) + " |")


if __name__ == "__main__":
    main()
