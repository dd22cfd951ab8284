"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs of bench.py into profiles/pmc_traffic.json + a markdown table.

usage: python tools/pmc_traffic.py <round tag> <N>:<fetch csv>:<write csv> [...]
HBM bytes per launch = 2 x FETCH_SIZE KB (gfx950 correction, MI355X_MICROARCH.md HBM section; calibrated on gf::action_kernel whose
byte count is known exactly) + WRITE_SIZE KB, median over the launches of the timed region."""
import csv
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALGO = {"post": 566, "action": 248, "synth_scene": 296}   # algorithmic B/env (scene: stand-in physics, informational)


def medians(path, counter):
    per = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            for key in ("post_ws_kernel", "post_kernel", "action_kernel", "synth_scene_kernel"):
                if key in name:
                    per.setdefault(name if key.startswith("post") else key, []).append(float(row["Counter_Value"]))
    return {k: statistics.median(v) for k, v in per.items() if len(v) >= 10}


def main():
    tag = sys.argv[1]
    out, lines = {}, []
    for spec in sys.argv[2:]:
        n, fpath, wpath = spec.split(":")
        n = int(n)
        fetch, write = medians(fpath, "FETCH_SIZE"), medians(wpath, "WRITE_SIZE")
        for name in sorted(fetch):
            if name not in write:
                continue
            kind = "post" if "post" in name else ("action" if "action" in name else "synth_scene")
            traffic = (2.0 * fetch[name] + write[name]) * 1024.0
            algo = ALGO[kind] * n
            short = name.replace("void ", "").split("(")[0]
            lines.append(f"| {n} | {short} | {fetch[name]:.1f} | {2 * fetch[name]:.1f} | {write[name]:.1f} | {traffic / 1024:.1f} | {algo / 1024:.1f} | {traffic / algo:.3f} |")
            if kind == "post":
                out[str(n)] = {"kernel": short, "fetch_size_kb": fetch[name], "write_size_kb": write[name], "traffic_bytes": traffic,
                               "algorithmic_bytes": algo, "round": tag}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print("| N | kernel | FETCH_SIZE KB (median) | corrected read KB | WRITE_SIZE KB (median) | traffic KB | algorithmic KB | traffic / algorithmic |")
    print("|---|---|---|---|---|---|---|---|")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
