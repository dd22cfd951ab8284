#!/bin/bash
# The round's evidence, collected on the GPU box into gpurun_out/ (copy what should be judged into profiles/):
#   tools/collect_profiles.sh <tag>
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/${tag}_bench_steps20.json 2>> gpurun_out/${tag}_bench.err
python3 tools/bench_configs.py > gpurun_out/${tag}_configs.jsonl 2> gpurun_out/${tag}_configs.err
GF_NO_CONTACT_FOLD=1 python3 tools/bench_configs.py --configs contacts,rough_terrain,humanoid,humanoid28,gait,gait_8192,gait_override_8192 > gpurun_out/${tag}_configs_nofold.jsonl 2>> gpurun_out/${tag}_configs.err
GF_OBS_OUTPUT=ring python3 tools/bench_configs.py --configs gait,gait_8192 > gpurun_out/${tag}_configs_ring.jsonl 2>> gpurun_out/${tag}_configs.err
GF_JIT=off python3 tools/bench_configs.py --configs go2_user > gpurun_out/${tag}_configs_nojit.jsonl 2>> gpurun_out/${tag}_configs.err
python3 tools/bench_configs.py --scene genesis_like --configs go2_cmd,go2_cmd_65536,rough_terrain,humanoid,gait,gait_8192 > gpurun_out/${tag}_configs_genesis_like.jsonl 2>> gpurun_out/${tag}_configs.err
python3 tools/bench_configs.py --scene genesis_like --no-trace --configs go2_cmd_65536,gait_8192 > gpurun_out/${tag}_configs_genesis_like_ordinary.jsonl 2>> gpurun_out/${tag}_configs.err
tools/microbench 1048576 > gpurun_out/${tag}_microbench_1m.txt 2>&1 || true
tools/microbench 65536 > gpurun_out/${tag}_microbench_65536.txt 2>&1 || true
python3 tools/bench_rollout.py 65536 > gpurun_out/${tag}_rollout.jsonl 2>> gpurun_out/${tag}_configs.err
{ for k in reward obs manager classes obsclass action curriculum anneal; do python3 tools/bench_user_term.py 65536 $k; done; python3 tools/bench_user_term.py 4096 manager; } > gpurun_out/${tag}_user_term.txt 2>> gpurun_out/${tag}_configs.err
{ python3 tools/bench_window.py 65536; python3 tools/bench_window.py 8192; } > gpurun_out/${tag}_window_output.jsonl 2>> gpurun_out/${tag}_configs.err
python3 tools/bench_reset_override.py > gpurun_out/${tag}_reset_override.txt 2>> gpurun_out/${tag}_configs.err
tools/prof_by_grid.sh ${tag}_bench bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-profile --no-sweep --no-hbm-point --no-pmc > /dev/null
tools/prof_by_grid.sh ${tag}_bench1m bench.py --steps 100 --warmup 10 --num-envs 1048576 --no-cpu-baseline --no-profile --no-sweep --no-hbm-point --no-pmc > /dev/null
tools/prof_by_grid.sh ${tag}_cfg tools/bench_configs.py > /dev/null
GF_NO_CONTACT_FOLD=1 tools/prof_by_grid.sh ${tag}_cfg_nofold tools/bench_configs.py --configs contacts,rough_terrain,humanoid,gait,gait_8192 > /dev/null
tools/scaling_table.sh ${tag} > /dev/null
for c in gait humanoid rough_terrain gait_8192 go2_cmd_65536; do python3 tools/stamp_config.py run $c 100 2>&1 | grep -v amdgpu.ids; done > gpurun_out/${tag}_stamps.txt || true
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$root/gpurun_out/${tag}_pmc_${c}_bench" -o pmc -- python3 "$root/bench.py" --steps 60 --warmup 10 --no-cpu-baseline --no-profile --no-sweep --no-hbm-point --no-pmc > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$root/gpurun_out/${tag}_pmc_${c}_bench1m" -o pmc -- python3 "$root/bench.py" --steps 40 --warmup 10 --num-envs 1048576 --no-cpu-baseline --no-profile --no-sweep --no-hbm-point --no-pmc > /dev/null 2>&1
  GF_JIT=off rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$root/gpurun_out/${tag}_pmc_${c}_cfg" -o pmc -- python3 "$root/tools/bench_configs.py" --steps 60 --configs gait,rough_terrain,humanoid,contacts > /dev/null 2>&1
  GF_JIT=off GF_NO_CONTACT_FOLD=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$root/gpurun_out/${tag}_pmc_${c}_cfg_nofold" -o pmc -- python3 "$root/tools/bench_configs.py" --steps 60 --configs gait,rough_terrain,humanoid,contacts > /dev/null 2>&1
done
cd "$root"
for k in bench bench1m cfg cfg_nofold; do
  python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_FETCH_SIZE_$k gpurun_out/${tag}_pmc_WRITE_SIZE_$k > gpurun_out/${tag}_pmc_$k.md
done
tail -c 400 gpurun_out/${tag}_bench.json; echo; cat gpurun_out/${tag}_pmc_bench.md | head -8
