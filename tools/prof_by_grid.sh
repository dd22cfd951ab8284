#!/bin/bash
# rocprofv3 kernel trace of a python command, summarised per (kernel, grid):   tools/prof_by_grid.sh <tag> <script> [args...]
# Writes gpurun_out/<tag>_by_grid.txt and gpurun_out/<tag>_kernel_stats.csv (copy what should be judged into profiles/).
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/${tag}_prof
rm -rf "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$tag" -- python3 "$root/$1" "${@:2}" > "$root/gpurun_out/${tag}.log" 2>&1
cd "$root"
python3 tools/rocprof_by_grid.py "$out" 5 > "gpurun_out/${tag}_by_grid.txt"
cp "$(find "$out" -name '*kernel_stats.csv' | head -1)" "gpurun_out/${tag}_kernel_stats.csv"
grep '^{' "gpurun_out/${tag}.log" || true
head -24 "gpurun_out/${tag}_by_grid.txt"
