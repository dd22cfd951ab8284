#!/bin/bash
# Step time of BASELINE configs 4 and 5 at the shard sizes of a strong-scaling run over 1 / 2 / 4 / 8 GPUs, measured on ONE GPU
# (a rank's step does not depend on the other ranks: no data-path collective), with the contact fold as shipped and switched off:
#   tools/scaling_table.sh <tag>
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
: > gpurun_out/${tag}_scaling.jsonl
for fold in default off; do
  if [ $fold = off ]; then export GF_NO_CONTACT_FOLD=1; else unset GF_NO_CONTACT_FOLD; fi
  for n in 65536 32768 16384 8192; do
    python3 tools/bench_configs.py --configs gait --num-envs $n 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); d['fold']='$fold'; print(json.dumps(d))" >> gpurun_out/${tag}_scaling.jsonl
  done
  for n in 8192 4096 2048 1024; do
    python3 tools/bench_configs.py --configs humanoid --num-envs $n 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); d['fold']='$fold'; print(json.dumps(d))" >> gpurun_out/${tag}_scaling.jsonl
  done
  for n in 16384 8192 4096 2048; do
    python3 tools/bench_configs.py --configs rough_terrain --num-envs $n 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); d['fold']='$fold'; print(json.dumps(d))" >> gpurun_out/${tag}_scaling.jsonl
  done
done
python3 -c "
import json
for l in open('gpurun_out/${tag}_scaling.jsonl'):
    d=json.loads(l); print(d['fold'], d['config'], d['num_envs'], round(d['us_per_step'],1))
"
