#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md's HBM
section prescribes): median over each kernel's launches, traffic = 2 x FETCH_SIZE KB (gfx950 correction) + WRITE_SIZE KB.
    python tools/pmc_summary.py <fetch dir-or-csv> <write dir-or-csv> [min launches]"""
import collections
import csv
import glob
import os
import statistics
import sys


def load(path, counter):
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            per[(r["Kernel_Name"].replace("void ", "").split("(")[0][:64], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))].append(float(r["Counter_Value"]))
    return per


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    print("| kernel | grid | launches | FETCH_SIZE KB | corrected read MB | WRITE_SIZE KB | traffic MB |")
    print("|---|---|---|---|---|---|---|")
    for key in sorted(f, key=lambda k: -sum(f[k])):
        if key not in w or len(f[key]) < min_n:
            continue
        fk, wk = statistics.median(f[key]), statistics.median(w[key])
        print(f"| {key[0]} | {key[1]} | {len(f[key])} | {fk:.1f} | {2 * fk / 1024:.2f} | {wk:.1f} | {(2 * fk + wk) / 1024:.2f} |")


if __name__ == "__main__":
    main()
