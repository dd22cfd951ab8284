#!/usr/bin/env python3
"""Print the per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite database or *_kernel_stats.csv)."""
import csv
import glob
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dbs = glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)
    csvs = glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)
    if csvs:
        rows = list(csv.DictReader(open(csvs[0])))
        for r in rows[:top]:
            print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"]) / 1e3:9.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:9.2f} {float(r["Percentage"]):5.1f}%')
        return
    if not dbs:
        raise SystemExit(f"no rocprofv3 output under {root}")
    cur = sqlite3.connect(dbs[0]).cursor()
    for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit ?", (top,)):
        print(f"{name[:70]:70s} calls {calls:6d} avg_us {avg / 1e3 if avg > 1e4 else avg:9.1f} total {total:12.1f} {pct:5.1f}%")


if __name__ == "__main__":
    main()
