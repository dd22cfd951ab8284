#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes, kernel trace only) of the configs with ContactManagers:
#   tools/pmc_cfg.sh <tag> [extra env assignments, e.g. GF_NO_CONTACT_FOLD=1]
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for kv in "$@"; do export "$kv"; done
export GF_JIT=off
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$root/gpurun_out/${tag}_pmc_${c}_cfg"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$root/gpurun_out/${tag}_pmc_${c}_cfg" -o pmc -- python3 "$root/tools/bench_configs.py" --steps 60 --configs gait,rough_terrain,humanoid,contacts > /dev/null 2>&1
done
cd "$root"
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_FETCH_SIZE_cfg gpurun_out/${tag}_pmc_WRITE_SIZE_cfg > gpurun_out/${tag}_pmc_cfg.md
grep "post_ws\|contact_kernel\|synth\|unroll" gpurun_out/${tag}_pmc_cfg.md
