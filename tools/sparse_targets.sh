#!/bin/bash
# VERDICT r3 #1's targets (humanoid 2 048 / 8 192 envs <= 19 us, gait 8 192 <= 30, gait 65 536 <= 100) were derived on round 3's near-empty
# contact tables.  This measures the configs on THAT workload (GF_SPARSE_CONTACTS=1) with the fold off / as shipped / forced:
#   tools/sparse_targets.sh <tag>
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
export GF_SPARSE_CONTACTS=1
out=gpurun_out/${tag}_sparse_targets.jsonl
: > $out
for fold in off default force; do
  unset GF_NO_CONTACT_FOLD GF_FORCE_CONTACT_FOLD
  [ $fold = off ] && export GF_NO_CONTACT_FOLD=1
  [ $fold = force ] && export GF_FORCE_CONTACT_FOLD=1
  for spec in humanoid:2048 humanoid:8192 gait:8192 gait:65536; do
    c=${spec%%:*}; n=${spec##*:}
    python3 tools/bench_configs.py --configs $c --num-envs $n 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); d['fold']='$fold'; d['contacts']='sparse (round 3)'; print(json.dumps(d))" >> $out
  done
done
python3 -c "
import json
for l in open('$out'):
    d=json.loads(l); print(d['fold'], d['config'], d['num_envs'], round(d['us_per_step'],1))
"
