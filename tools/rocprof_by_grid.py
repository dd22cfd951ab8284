#!/usr/bin/env python3
"""Per (kernel, grid size) launch count and average duration from a rocprofv3 --kernel-trace --output-format csv run."""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    paths = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not paths:
        raise SystemExit(f"no *kernel_trace.csv under {root}")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(paths[0])):
        agg[(r["Kernel_Name"][:56], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for (name, grid, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if len(v) >= min_calls:
            v.sort()
            print(f"{name:56s} grid {grid:9d} wg {wg:4d} calls {len(v):6d} avg_us {sum(v) / len(v):8.1f} med_us {v[len(v) // 2]:8.1f}")


if __name__ == "__main__":
    main()
