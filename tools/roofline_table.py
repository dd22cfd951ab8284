#!/usr/bin/env python3
"""Per-kernel table out of a collected profile set: rocprofv3 duration (by-grid summary) x PMC traffic (FETCH_SIZE / WRITE_SIZE passes)
-> bytes moved per second and the fraction of the 8 TB/s HBM3E peak, for every library kernel of the bench / bench-at-1M / config runs.
    python tools/roofline_table.py profiles/r03_final > profiles/r03_final_roofline_by_kernel.md
Traffic is what the memory system MOVED (measured), not the algorithmic bytes: DESIGN.md states those per kernel; where the two
differ (contact launch, array-of-structs gathers) the table's fraction is the generous one."""
import re
import sys

PEAK = 8000.0   # GB/s


def by_grid(path):
    out = {}
    for line in open(path):
        m = re.match(r"(.*?)\s+grid\s+(\d+)\s+wg\s+(\d+)\s+calls\s+(\d+)\s+avg_us\s+([\d.]+)\s+med_us\s+([\d.]+)", line)
        if m:
            name = re.sub(r"^void ", "", m.group(1)).split("(")[0].strip()
            out[(name, int(m.group(2)))] = (float(m.group(5)), float(m.group(6)), int(m.group(4)))
    return out


def pmc(path):
    out = {}
    for line in open(path):
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) == 7 and c[1].isdigit():
            out[(c[0], int(c[1]))] = float(c[6])
    return out


def main(prefix):
    print("| run | kernel | grid (threads) | launches | median µs | PMC traffic MB | TB/s | of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|")
    for run in ("bench", "bench1m", "cfg"):
        try:
            g, p = by_grid(f"{prefix}_{run}_by_grid.txt"), pmc(f"{prefix}_pmc_{run}.md")
        except FileNotFoundError:
            continue
        rows = []
        for (name, grid), mb in p.items():
            hit = [(k, v) for k, v in g.items() if k[1] == grid and (k[0].startswith(name[:40]) or name.startswith(k[0][:40]))]
            if not hit or not name.startswith("gf::"):
                continue
            (kname, _), (avg, med, calls) = hit[0]
            tbps = mb / med   # MB per µs = TB/s
            rows.append((med, f"| {run} | `{name}` | {grid} | {calls} | {med:.1f} | {mb:.2f} | {tbps:.2f} | {tbps * 1000 / PEAK:.2f} |"))
        for _, r in sorted(rows, reverse=True):
            print(r)


if __name__ == "__main__":
    main(sys.argv[1])
