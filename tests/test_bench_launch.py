"""bench.py --gpus N starts its ranks itself: rehearsed here with 2 ranks over gloo on the CPU.

The children are ordinary `python bench.py …` processes; tests/inject/sitecustomize.py (on their PYTHONPATH) installs the oracle
as the compute backend before bench.py runs, so everything except the kernels is the real thing: the parent that never touches
the GPU, the rendezvous environment, one process group, contiguous env shards, the per-batch max over ranks, one JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=os.path.join(ROOT, "tests", "inject") + os.pathsep + env.get("PYTHONPATH", ""), GF_TEST_INJECT_ORACLE="1",
               GF_DEVICE="cpu", OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]   # exactly one line: libraries that greet on stdout (gloo, RCCL) are sent to stderr
    assert len(lines) == 1, f"exactly one JSON line expected, got {len(lines)}: {p.stdout[-500:]}"
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_gpus_2_self_launch_weak():
    out = _bench("--gpus", "2", "--steps", "6", "--warmup", "2", "--num-envs", "96", "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 6
    assert out["config"]["num_envs_per_gpu"] == 96 and out["config"]["global_num_envs"] == 192
    assert out["config"]["parallelism"] == "env-shard x2" and out["config"]["dist_backend"] == "gloo"
    assert out["value"] > 0 and abs(out["value"] - 192 * 6 / (out["ms_per_step"] * 6e-3)) < 1e-6 * out["value"]
    assert out["timing"]["batches"] >= 1 and out["timing"]["timed_s"] >= 0.3
    # every rank pins itself to its own share of the host cores (from inside the rank) and reports it, with the host's
    # own time per step; the IPC mode RCCL needs on this driver is reported and comes from the environment when that sets it
    ranks = out["ranks"]
    assert all(r["host_us_per_step"] > 0 for r in ranks)
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 4:
        assert all(r["affinity"]["pinned"] and r["affinity"]["cpus"] == ncpu // 2 for r in ranks)
        assert ranks[0]["affinity"]["last"] < ranks[1]["affinity"]["first"], "disjoint core sets"
    assert out["config"]["hsa_enable_ipc_mode_legacy"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


@pytest.mark.timeout(300)
@pytest.mark.parametrize("config,global_envs", [("gait", 140), ("humanoid", 132)])
def test_gpus_2_strong_scaling_with_a_collective_per_step(config, global_envs):
    """--scaling strong --config gait|humanoid with --reduce-every 1: the K = 1 path (one all-reduce of one statistics row per step,
    rank-local log reads) next to the K = 32 default of the other cases; GF_PIN=0 leaves the affinity alone."""
    os.environ["GF_PIN"] = "0"
    try:
        out = _bench("--gpus", "2", "--steps", "5", "--warmup", "2", "--scaling", "strong", "--global-envs", str(global_envs), "--config", config,
                     "--reduce-every", "1", "--no-cpu-baseline")
    finally:
        del os.environ["GF_PIN"]
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["workload"] == config
    assert out["config"]["stats_allreduce_every_steps"] == 1 and out["config"]["global_num_envs"] == global_envs
    assert [r["num_envs"] for r in out["ranks"]] == [global_envs // 2] * 2 and not any(r["affinity"]["pinned"] for r in out["ranks"])
    assert out["value"] > 0


@pytest.mark.timeout(300)
def test_gpus_2_self_launch_strong_gait():
    """BASELINE config 5's shape: a global env count split over the ranks (65 536 over 8 on the real node; 130 over 2 here)."""
    out = _bench("--gpus", "2", "--steps", "5", "--warmup", "2", "--scaling", "strong", "--global-envs", "130", "--config", "gait", "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["global_num_envs"] == 130 and out["config"]["num_envs_per_gpu"] == 65 and out["config"]["workload"] == "gait"
    assert out["value"] > 0


@pytest.mark.timeout(420)
def test_gpus_8_self_launch_strong_gait():
    """The driver's scaling run in small: 8 ranks, BASELINE config 5 (gait), a global env count split into contiguous shards, the
    statistics ring reduced 32 steps at a time, one JSON line whose per-rank entries come from the process group."""
    out = _bench("--gpus", "8", "--steps", "40", "--warmup", "2", "--scaling", "strong", "--global-envs", "520", "--config", "gait", "--no-cpu-baseline",
                 timeout=400)
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8 and out["scaling"] == "strong"
    assert out["config"]["global_num_envs"] == 520 and out["config"]["num_envs_per_gpu"] == 65 and out["config"]["stats_allreduce_every_steps"] == 32
    ranks = out["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8))
    assert [r["env_offset"] for r in ranks] == [65 * r for r in range(8)] and all(r["num_envs"] == 65 for r in ranks)
    assert all(r["batch_ms_median"] > 0 for r in ranks)
    assert out["ms_per_step"] * 40 >= max(r["batch_ms_median"] for r in ranks) * 0.5   # the reported time is the max over ranks, per batch
    assert out["value"] > 0


@pytest.mark.timeout(300)
def test_single_rank_line_and_torchrun_env():
    """N = 1 is unchanged (no process group), and ranks started by an external launcher (WORLD_SIZE set) do not self-launch."""
    out = _bench("--gpus", "1", "--steps", "4", "--warmup", "1", "--num-envs", "64", "--no-cpu-baseline")
    assert out["n_gpus"] == 1 and out["config"]["parallelism"] == "env-shard x1" and out["config"]["dist_backend"] is None
    assert out["rccl_ranks"] == 1 and len(out["ranks"]) == 1 and out["ranks"][0]["num_envs"] == 64


def test_spawn_ranks_propagates_failure_and_environment(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
    from genesis_forge_amd.launch import spawn_ranks

    code = ("import os, sys; open(os.path.join(sys.argv[1], 'r' + os.environ['RANK']), 'w').write(' '.join(os.environ[k] for k in "
            "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR'))); import time; time.sleep(1.5 if os.environ['RANK'] == '1' else 0); "
            "sys.exit(3 if os.environ['RANK'] == '1' else 0)")
    rc = spawn_ranks([sys.executable, "-c", code, str(tmp_path)], 3)
    assert rc == 3
    for r in range(3):
        assert (tmp_path / f"r{r}").read_text() == f"{r} {r} 3 127.0.0.1"


def test_rank_pinning_plan_is_numa_local_when_the_box_says_so():
    """bench.py:pin_plan — on an 8-GPU node (two sockets, SMT siblings numbered behind the physical cores) a contiguous split would give
    ranks 2 and 3 the other socket's cores and make ranks 0 and 4 SMT siblings; with the GPUs' NUMA nodes from sysfs the ranks of a socket
    split that socket's cores.  Without the information (or with an unusable node) it is the contiguous split."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    avail = list(range(256))
    node_cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    gpu_nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    plans = [bench.pin_plan(r, 8, avail, gpu_nodes, node_cpus) for r in range(8)]
    assert all(len(p) == 32 for p in plans) and len(set(c for p in plans for c in p)) == 256   # disjoint, everything used
    assert all(set(plans[r]) <= set(node_cpus[0]) for r in range(4)) and all(set(plans[r]) <= set(node_cpus[1]) for r in range(4, 8))
    assert [bench.pin_plan(r, 8, avail, [], {}) for r in range(8)] == [list(range(32 * r, 32 * r + 32)) for r in range(8)]
    assert bench.pin_plan(3, 8, avail, [0, 0, 0, -1, 1, 1, 1, 1], node_cpus) == list(range(96, 128))   # this rank's GPU unknown: its flat share
    assert bench.pin_plan(1, 2, list(range(8)), [5, 5], {5: [100, 101]}) == [4, 5, 6, 7]              # the node's cores are not ours to use
    assert bench._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
