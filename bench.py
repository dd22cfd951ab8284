#!/usr/bin/env python3
"""
bench.py — env-steps/sec of the ManagedEnvironment manager-step pipeline on MI355X.

A "step" is one full ManagedEnvironment.step() (action → synthetic scene tick → termination → reward → command → masked
reset → observation) over one batch of envs resident in HBM: BASELINE.json's Go2 12-DOF config with the full Reward /
Termination / Command manager stack (6 reward terms, 2 termination terms, velocity command, 48-wide observation), synthetic
data.  The path shards by env: every rank owns its own envs, the only cross-rank traffic is the logging all-reduce (RCCL).

    python bench.py                                   # 1 GPU, 65 536 envs
    python bench.py --gpus 8                          # starts 8 ranks itself (one per GPU), weak scaling: 65 536 envs per GPU
    python bench.py --gpus 8 --scaling strong --global-envs 65536 --config gait     # BASELINE config 5 sharded over 8 GPUs
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Timing: W warm-up steps, then BATCHES of exactly K steps, each bracketed by barrier + synchronize on both sides and taken as
the max over ranks, repeated until at least 0.3 s has been timed; the MEDIAN batch is reported (``ms_per_step`` = median / K).
Nothing else runs inside a timed batch: the HIP-event stamping of the dominant kernel (``roofline``) is a separate loop after
the timed batches, and every figure in the JSON line is measured by this run (committed rocprofv3 / PMC summaries live under
profiles/ and are quoted in DESIGN.md, not here).

Prints ONE JSON line (rank 0).  ``roofline``: the fused post-physics launch that contains the reward fold (566 algorithmic
B/env for this config), bytes ÷ its average duration from dispatch-timestamp HIP events on the launch stream.
``roofline_hbm``: the same kernel measured in this run at 1 048 576 envs, where the working set (0.85 GB) no longer fits the
256 MiB Infinity Cache.  ``cpu_baseline``: the CPU oracle (oracle/, a scalar C port of the reference algorithm) on the host
cores of the same box over the N x D grid of BASELINE.md §3 (plan B2).
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
MIN_TIMED_S = 0.3
PRIMING_STEPS = 8      # setup, not warm-up: an env records its step over its first two steps and the first replay loads the
#                        fused kernel's code object — done before the W warm-up steps so that even `--warmup 0` times the steady state
HBM_POINT_ENVS = 1 << 20
GRID_N = (64, 4096, 16384, 65536)
GRID_D = (12, 28)
# the package's default output contract (the reference's: every step returns a tensor of the caller's own — the fused launch writes
# straight into it); GF_OBS_OUTPUT=static selects the persistent output slots instead (same step time within the box-to-box noise)
OBS_OUTPUT = os.environ.get("GF_OBS_OUTPUT", "fresh")


def reward_bytes_per_env(D: int, T: int, cmd_width: int) -> int:
    """Algorithmic bytes of one reward-kernel launch per env: every input read once, every output
    written once: pos 12 + quat 16 + vel 12 + ang 12 + 3 [N,D] rows + command, RW episode sums 8T, RW seconds 8, W reward 4."""
    return 4 * (13 + 3 * D + cmd_width) + 8 * T + 12


def post_bytes_per_env(D: int, T: int, cmd_width: int, O: int, H: int = 1) -> int:
    """Algorithmic bytes per env of the fused post-physics launch for this config (SURVEY.md §8d "fully fused post-physics
    kernel"): read pos/quat/vel/ang 52, five [N,D] rows (dof_pos, dof_vel, targets, actions, last_actions), the command row,
    episode_length + max_episode_length 8, episode_seconds 4, T episode sums; write 2 masks, reward 4, T sums, seconds 4 and the
    O*H observation (plus the (H-1) history frames it re-reads).  = 566 for D=12, T=6, R=3, O=48, H=1."""
    reads = 52 + 5 * 4 * D + 4 * cmd_width + 8 + 4 + 4 * T + 4 * O * (H - 1)
    writes = 2 + 4 + 4 * T + 4 + 4 * O * H
    return reads + writes


def pmc_traffic(num_envs: int, kernel_substr: str = "post_ws_kernel", timeout_s: float = 150.0):
    """HBM traffic per launch of the dominant kernel from the PMC counters, measured NOW: two child runs of this very file (short,
    nothing but the recorded step) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` — separate passes, no tracing domain
    besides the kernel trace, as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes — and its gfx950 correction: FETCH_SIZE
    reports half of the bytes of wide coalesced reads, WRITE_SIZE is exact; both in KB.  traffic = 2 * FETCH_SIZE + WRITE_SIZE,
    median over the kernel's launches.  Returns (bytes per launch, details) or (None, reason): a missing profiler, a refused
    counter or a timeout leaves `traffic` null, it never costs the bench line."""
    import csv
    import glob
    import shutil
    import tempfile

    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found"
    med = {}
    work = tempfile.mkdtemp(prefix="gf_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "40", "--warmup", "5", "--num-envs", str(num_envs),
                   "--no-cpu-baseline", "--no-profile", "--no-sweep", "--no-hbm-point", "--no-pmc"]
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 --pmc {counter} timed out after {timeout_s:.0f} s"
            if p.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} exited with {p.returncode}: {p.stderr[-200:]}"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"rocprofv3 --pmc {counter} wrote no counter_collection.csv"
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                    if r.get("Counter_Name") == counter and kernel_substr in r.get("Kernel_Name", "")]
            if len(vals) < 10:
                return None, f"only {len(vals)} launches of {kernel_substr} in the {counter} pass"
            med[counter] = (statistics.median(vals), len(vals))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    fetch_kb, write_kb = med["FETCH_SIZE"][0], med["WRITE_SIZE"][0]
    traffic = (2.0 * fetch_kb + write_kb) * 1024.0
    return traffic, {"FETCH_SIZE_KB_median": fetch_kb, "WRITE_SIZE_KB_median": write_kb, "launches": [med["FETCH_SIZE"][1], med["WRITE_SIZE"][1]],
                     "rule": "traffic = 2 x FETCH_SIZE (gfx950: the counter tallies 128-B read requests at 64 B) + WRITE_SIZE, KB -> bytes, "
                             "median over the kernel's launches of two separate rocprofv3 --pmc child runs of this file"}


def make_env(num_envs: int, config: str = "go2_cmd", dofs: int = 12):
    from genesis_forge_amd import tasks
    from genesis_forge_amd.managers import ObservationManager

    # the mode the step's observations are returned in: the package default "fresh" — the reference's contract, a new tensor per
    # call that the step's launch writes directly — unless GF_OBS_OUTPUT asks for "static" (persistent slots) or "ring"
    ObservationManager.default_output = OBS_OUTPUT

    if config == "go2_cmd":
        env = tasks.bench_env(num_envs, dofs=dofs)
    else:
        env = tasks.BASELINE_CONFIGS[config][1](num_envs)
    env.build()
    return env


# ------------------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = "port"; BASELINE.md §3 plan B2): N x D grid, one thread and all host cores
# ------------------------------------------------------------------------------------------------------------------------------
def _oracle_env(num_envs: int, dofs: int):
    """The bench workload on CPU tensors with the oracle as the compute backend — the checker timed as the CPU baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend

    if not isinstance(nat._backend, OracleBackend):
        gs.set_device("cpu")
        nat.set_backend(OracleBackend(os.path.join(ROOT, "oracle", "libgf_oracle.so")))
    env = make_env(num_envs, dofs=dofs)
    env.seed(1234)
    env.reset()
    g = torch.Generator().manual_seed(0)
    acts = [torch.randn(num_envs, dofs, generator=g) for _ in range(4)]
    for i in range(3):
        env.step(acts[i % 4])
    return env, acts


def _cpu_point(env, acts, num_envs, min_iters, budget_s):
    """Step until `min_iters` iterations AND `budget_s` seconds have passed (at most 4 x budget); median iteration time."""
    ts = []
    t_end = time.perf_counter() + budget_s
    hard = time.perf_counter() + 4 * budget_s + 3.0
    while (len(ts) < min_iters or time.perf_counter() < t_end) and time.perf_counter() < hard:
        t0 = time.perf_counter()
        env.step(acts[len(ts) % 4])
        ts.append(time.perf_counter() - t0)
    med = statistics.median(ts)
    return {"iters": len(ts), "us_per_step_median": med * 1e6, "value": num_envs / med}


def cpu_worker(shard_div: int, budget_s: float) -> None:
    """One host core: builds every grid point at num_envs / shard_div envs, reports "ready", waits for "go" on stdin (so all
    workers measure the same point at the same time), then prints one JSON line with its per-point throughput."""
    import torch

    torch.set_num_threads(1)
    points = []
    for d in GRID_D:
        for n in GRID_N:
            shard = max(1, n // shard_div)
            points.append((n, d, shard) + _oracle_env(shard, d))
    print("ready", flush=True)
    sys.stdin.readline()
    out = []
    for n, d, shard, env, acts in points:
        t_end = time.perf_counter() + budget_s
        steps = 0
        t0 = time.perf_counter()
        while time.perf_counter() < t_end:
            env.step(acts[steps % 4])
            steps += 1
        out.append({"num_envs": n, "dofs": d, "value": shard * steps / (time.perf_counter() - t0)})
    print(json.dumps(out), flush=True)


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(headline_envs: int, workers: int) -> dict:
    """Runs in helper processes that never open the GPU (GF_DEVICE=cpu, no visible devices)."""
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", GF_DEVICE="cpu", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="",
               CUDA_VISIBLE_DEVICES="")
    me = os.path.abspath(__file__)
    # (1) one thread: median of >= 200 iterations per grid point
    single = subprocess.run([sys.executable, me, "--cpu-single"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600)
    grid = json.loads(single.stdout.strip().splitlines()[-1])
    # (2) all host cores of this GPU's share: the path shards by env on the CPU exactly as across GPUs — `workers` processes, each
    # stepping num_envs / workers envs with the single-threaded oracle; the figure is the sum over workers
    multi = None
    if workers > 1:
        procs = [subprocess.Popen([sys.executable, me, "--cpu-worker", str(workers), "0.8"], env=env, stdin=subprocess.PIPE,
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(workers)]
        try:
            for p in procs:
                assert p.stdout.readline().strip() == "ready"
            for p in procs:
                p.stdin.write("go\n")
                p.stdin.flush()
            rows = [json.loads(p.stdout.readline()) for p in procs]
            for p in procs:
                p.wait(timeout=60)
            multi = [{"num_envs": r0["num_envs"], "dofs": r0["dofs"], "cores": workers, "value": sum(r[i]["value"] for r in rows)}
                     for i, r0 in enumerate(rows[0])]
        except Exception:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    head = next(g for g in grid if g["num_envs"] == (headline_envs if headline_envs in GRID_N else 65536) and g["dofs"] == 12)
    return {"value": head["value"], "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{head['iters']} steps x {head['num_envs']} envs of the same Go2 full-stack workload (median step), oracle/libgf_oracle.so, one thread",
            "cpu_model": _cpu_model(), "nproc": os.cpu_count(),
            "grid": [dict(g, cores=1, unit="env-steps/s") for g in grid],
            "grid_multicore": multi,
            "grid_note": "BASELINE.md plan B2: N in {64, 4096, 16384, 65536} x D in {12, 28}; D = 28 is the same manager stack over a synthetic "
                         "28-joint robot (O = 96); one thread = median of >= 200 steps; multicore = sum over worker processes each stepping "
                         "N / workers envs for 0.8 s, all workers on the same grid point at the same time",
            # not measured by this run — quoted so the line carries it: the REFERENCE's own PyTorch managers cannot travel to the GPU box
            "reference_quoted": {"value": 5.50e6, "unit": "env-steps/s", "cores": 8, "num_envs": 65536,
                                 "what": "genesis-forge's own ManagedEnvironment.step (PyTorch CPU, stub physics), 8-core Xeon 2.1 GHz build container",
                                 "source": "BASELINE.md section 2 (survey probe); one thread: 2.07e6"}}


def cpu_single() -> None:
    import torch

    torch.set_num_threads(1)
    out = []
    for d in GRID_D:
        for n in GRID_N:
            env, acts = _oracle_env(n, d)
            out.append(dict({"num_envs": n, "dofs": d}, **_cpu_point(env, acts, n, 200, 0.5)))
            del env
    print(json.dumps(out), flush=True)


# ------------------------------------------------------------------------------------------------------------------------------
def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node; > 1 without WORLD_SIZE in the environment: bench.py starts the ranks itself")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --num-envs envs per GPU; strong: --global-envs envs in total, sharded contiguously over the GPUs")
    ap.add_argument("--num-envs", type=int, default=65536, help="envs per GPU (weak scaling)")
    ap.add_argument("--global-envs", type=int, default=65536, help="total envs (strong scaling)")
    ap.add_argument("--config", default="go2_cmd", help="workload: go2_cmd (BASELINE.json's headline config) or another key of "
                                                        "genesis_forge_amd.tasks.BASELINE_CONFIGS (gait = config 5, humanoid = config 4)")
    ap.add_argument("--reduce-every", type=int, default=0,
                    help="recorded steps per logging all-reduce (world > 1): the statistics rows of K consecutive steps travel as ONE all-reduce "
                         "(K x 392 B).  0 (default) = 32 — the benchmark loop is lock-step on every rank, which is what K > 1 asks for "
                         "(distributed.attach); 1 = one asynchronous all-reduce per step, log reads rank-local (the library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the 4096 / 16384-env side measurements")
    ap.add_argument("--no-hbm-point", action="store_true", help="skip the 1 048 576-env roofline_hbm measurement")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event stamping pass (roofline)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip roofline.traffic: two short child runs of this file under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` "
                         "(separate passes, as MI355X_MICROARCH.md's HBM section prescribes) after everything else has been measured")
    ap.add_argument("--profile-samples", type=int, default=200, help="stamped launches of the dominant kernel in the stamping pass")
    ap.add_argument("--profile-stride", type=int, default=4,
                    help="the stamping pass stamps every k-th launch: a stamped launch blocks the host for ~12 us and drains the queue, "
                         "so the launches between two stamps let the pipeline refill")
    ap.add_argument("--profile-presleep-cycles", type=int, default=150000,
                    help="GPU spin (torch.cuda._sleep) enqueued in front of every stamped step of the stamping pass so that the stamped "
                         "kernel runs behind queued work as in the timed loop; 0 = off")
    return ap.parse_args(argv)


def main():
    argv = sys.argv[1:]
    if argv and argv[0] == "--cpu-worker":
        return cpu_worker(int(argv[1]), float(argv[2]))
    if argv and argv[0] == "--cpu-single":
        return cpu_single()
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Start one rank per GPU ourselves.  Nothing in this process touches the GPU: launch.py is loaded from its FILE (importing
        # the package would run gs._default_device(), i.e. torch.cuda.is_available(), which opens the runtime) and needs only the
        # standard library; the ranks are fresh interpreters, this one only relays rank 0's line and exits with their status.
        import importlib.util

        spec = importlib.util.spec_from_file_location(
            "_gf_launch", os.path.join(ROOT, "genesis-forge_amd", "genesis_forge_amd", "launch.py"))
        launch = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(launch)
        sys.exit(launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + argv, args.gpus))
    run_rank(args)


def _cpulist(text: str) -> list:
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes() -> list:
    """NUMA node of every GPU of this box, in KFD topology order (the order HIP enumerates devices in), read from sysfs — no HIP call:
    /sys/class/kfd/kfd/topology/nodes/<n>/properties (simd_count > 0: a GPU; drm_render_minor) →
    /sys/class/drm/renderD<minor>/device/numa_node.  [] when the box does not say."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    out = []
    try:
        for n in sorted(os.listdir(base), key=int):
            props = dict(line.split()[:2] for line in open(os.path.join(base, n, "properties")) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
            node = -1
            try:
                node = int(open(f"/sys/class/drm/renderD{minor}/device/numa_node").read())
            except (OSError, ValueError):
                pass
            out.append(node)
    except (OSError, ValueError):
        return []
    return out


def pin_plan(local: int, local_world: int, avail: list, gpu_nodes: list, node_cpus: dict) -> list:
    """The cores rank `local` of `local_world` runs on.  When the box says which NUMA node every rank's GPU hangs off, the ranks whose
    GPUs share a node split THAT node's cores (of the ones this process may use) among themselves, in rank order; otherwise — or when a
    node's share would be empty — the available cores are split into `local_world` contiguous shares."""
    share = len(avail) // local_world
    flat = avail[local * share:(local + 1) * share]
    if len(gpu_nodes) < local_world or gpu_nodes[local] < 0:
        return flat
    node = gpu_nodes[local]
    peers = [r for r in range(local_world) if gpu_nodes[r] == node]
    cpus = [c for c in node_cpus.get(node, []) if c in set(avail)]
    per = len(cpus) // len(peers)
    if per < 1:
        return flat
    k = peers.index(local)
    return cpus[k * per:(k + 1) * per]


def pin_rank() -> dict:
    """Give this rank its own share of the host cores (``os.sched_setaffinity``, from inside the rank): below ~16 k envs a step is
    host-bound (17 us at 4 096 envs), so N unpinned Python hosts on one box would migrate over each other's cores.  The share is
    NUMA-local to the rank's GPU when sysfs says where the GPUs are (``pin_plan``); ``GF_PIN=0`` leaves the affinity alone.  Returns
    what was done, for the per-rank entry of the JSON line."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:   # not Linux
        return {"pinned": False, "cpus": None, "why": "no sched_getaffinity"}
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))) % max(1, local_world)
    if os.environ.get("GF_PIN", "1") == "0" or local_world <= 1 or len(avail) < 2 * local_world:
        return {"pinned": False, "cpus": len(avail), "why": "GF_PIN=0" if os.environ.get("GF_PIN", "1") == "0" else "one rank, or fewer than two cores per rank"}
    nodes = gpu_numa_nodes()
    node_cpus = {}
    for n in set(x for x in nodes if x >= 0):
        try:
            node_cpus[n] = _cpulist(open(f"/sys/devices/system/node/node{n}/cpulist").read())
        except OSError:
            pass
    mine = pin_plan(local, local_world, avail, nodes, node_cpus)
    try:
        os.sched_setaffinity(0, mine)
    except OSError as e:
        return {"pinned": False, "cpus": len(avail), "why": repr(e)}
    return {"pinned": True, "cpus": len(mine), "first": mine[0], "last": mine[-1],
            "numa_node": (nodes[local] if len(nodes) >= local_world and nodes[local] >= 0 and mine and mine[0] in node_cpus.get(nodes[local], []) else None)}


def run_rank(args):
    # stdout carries exactly ONE line, the JSON result of rank 0.  Libraries write there too (RCCL prints a five-line version banner
    # on rank 0 when the first communicator is created; gloo a connection note), so the descriptor is kept aside and fd 1 points at
    # stderr while the rank runs.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        affinity = pin_rank()   # BEFORE anything touches the GPU (and never through taskset / numactl: under a profiler that is a forbidden exec)
    except Exception as e:      # (pinning is an optimisation: an unexpected sysfs layout on a node nobody has seen must not stop the run)
        affinity = {"pinned": False, "cpus": None, "why": repr(e)}

    import torch
    import torch.distributed as dist
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import distributed as gfd
    from genesis_forge_amd import gs

    # A test harness may have injected a CPU backend (tests/test_bench_launch.py rehearses the multi-rank path over gloo with
    # the oracle); bench.py itself never does: without it the manager phases only exist as HIP kernels.
    rehearsal = nat._backend is not None and nat._backend.device_type == "cpu"
    if not rehearsal and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the manager phases only exist as HIP kernels")
    # one rank per GPU over RCCL ("nccl" on ROCm).  GF_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer
    # GPUs than ranks (ranks then share devices; RCCL refuses that)
    dist_backend = os.environ.get("GF_DIST_BACKEND", "gloo" if rehearsal else "nccl")
    if rehearsal:
        gs.set_device("cpu")
    else:
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        os.environ["LOCAL_RANK"] = str(local) if dist_backend != "nccl" else os.environ.get("LOCAL_RANK", "0")
        gs.set_device(f"cuda:{local}")
    rank, world = gfd.init_from_env(dist_backend)
    # GF_DIST_FORCE=1: go through the process-group code (RCCL barrier, max over ranks, batched statistics all-reduce) even with one
    # rank — how the multi-GPU path of this file is rehearsed on a one-GPU box; the numbers are then those of one GPU plus collectives
    dist_on = world > 1
    if world == 1 and os.environ.get("GF_DIST_FORCE") == "1" and not rehearsal:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29631")
        dist.init_process_group(dist_backend, rank=0, world_size=1)
        dist_on = True
    if dist_on:
        assert dist.get_world_size() == world
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    backend = nat.get_backend()
    sync = (lambda: None) if rehearsal else torch.cuda.synchronize

    if args.reduce_every <= 0:
        args.reduce_every = 32
    if args.scaling == "strong":
        start, N = gfd.shard(args.global_envs, rank, world)
        global_envs = args.global_envs
    else:
        N, global_envs = args.num_envs, args.num_envs * world
    env = make_env(N, args.config)
    D = env.action_space.shape[0]
    gfd.attach(env, global_num_envs=global_envs, reduce_every=args.reduce_every, force=dist_on, lockstep_reads=True)   # (every rank reads the same logs)
    env.seed(1234)          # one seed for all ranks: Philox is keyed by the GLOBAL env id, so shards draw different numbers
    env.reset()
    g = torch.Generator().manual_seed(1234 + rank)
    acts = [torch.randn(N, D, generator=g).to(gs.device) for _ in range(8)]

    local_times: list = []
    host_times: list = []   # the step loop's own time, before the final synchronize: what the HOST spends per batch

    def barrier():
        sync()
        if dist_on:
            dist.barrier()
        sync()

    def timed_batches(e, a, steps):
        """Batches of exactly `steps` steps, barrier + synchronize on both sides, max over ranks, until MIN_TIMED_S is covered."""
        times = []
        while True:
            barrier()
            t0 = time.perf_counter()
            for i in range(steps):
                e.step(a[i % len(a)])
            host_dt = time.perf_counter() - t0
            sync()
            dt = time.perf_counter() - t0
            barrier()
            local_times.append(dt)   # this rank's own clock, before the max over ranks (reported per rank)
            host_times.append(host_dt / steps)
            if dist_on:
                t = torch.tensor([dt], device=gs.device, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            times.append(dt)
            if sum(times) >= MIN_TIMED_S or len(times) >= 1000:
                return times

    def stamp_pass(e, a, bytes_per_launch):
        """HIP-event (dispatch timestamp) durations of the dominant kernel, in a loop of its own after the timed batches."""
        fused_ = e._trace is not None and e._trace.post_refs is not None
        phase = nat.GF_PHASE_POST if fused_ else nat.GF_PHASE_REWARD
        stride = max(1, args.profile_stride)
        backend.set_option(nat.GF_OPT_PROFILE_STRIDE, stride)
        backend.profile_begin(phase, args.profile_samples)
        for i in range(stride * args.profile_samples):
            if i % stride == 0 and args.profile_presleep_cycles > 0:
                # A stamped launch blocks the host for ~12 us; at 65 536 envs the GPU would drain its queue meanwhile and the stamped
                # kernel would start on an idle GPU (its dispatch then includes the launch ramp that back-to-back kernels hide).
                # A spin kernel in front of the stamped step keeps the stream busy while the host enqueues it, so the step's three
                # kernels run back to back exactly as in the timed loop.  It touches no memory (caches are as the step left them).
                torch.cuda._sleep(args.profile_presleep_cycles)
            e.step(a[i % len(a)])
        sync()
        ms, cnt = backend.profile_end()
        if cnt <= 0:
            return None
        avg_s = ms / cnt / 1e3
        achieved = bytes_per_launch / avg_s / 1e9
        return {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "avg_launch_us": avg_s * 1e6, "launches": cnt}

    for i in range(PRIMING_STEPS):
        env.step(acts[i % 8])
    for i in range(args.warmup):
        env.step(acts[i % 8])
    times = timed_batches(env, acts, args.steps)
    batch = statistics.median(times)
    # what each rank ran on and measured by its own clock: makes the one line checkable against the launcher's view of the node
    mine = {"rank": rank, "device": ("cpu (rehearsal)" if rehearsal else torch.cuda.get_device_name(gs.device)), "device_index": (None if rehearsal else gs.device.index),
            "num_envs": N, "env_offset": env.env_offset, "batch_ms_median": statistics.median(local_times) * 1e3, "batch_ms_max": max(local_times) * 1e3,
            "host_us_per_step": statistics.median(host_times) * 1e6, "affinity": affinity}
    per_rank = [mine]
    if dist_on:
        per_rank = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank, mine)
    _ = dict(env.extras["episode"])   # keep the logging path honest: read one step's global statistics (on every rank)
    tm_ = env.managers["termination"]
    reset_frac = float((tm_._terminated_buf | tm_._truncated_buf).float().mean()) if tm_ is not None else 0.0   # (after the timed region)

    fused = env._trace is not None and env._trace.post_refs is not None
    rm = env.managers["reward"]
    T = sum(1 for c in rm.cfg.values() if c.weight != 0)
    go2 = args.config == "go2_cmd"
    O = 12 + 3 * D
    per_env = (post_bytes_per_env(D, T, 3, O, 1) if fused else reward_bytes_per_env(D, T, 3)) if go2 else None
    kernel = "gf::reward_kernel"
    if fused and not rehearsal:
        what = backend.post_describe(env._trace.post_refs).split(":")[0]   # "program <id> (<name>)"
        kernel = f"gf::post_ws_kernel, {what} (termination+reward+command+reset+observe fused)"
    roof = None
    if not args.no_profile and not rehearsal and per_env is not None:
        m = stamp_pass(env, acts, per_env * N)   # on every rank (the steps contain the logging collective); rank 0 reports its own
        if m is not None:
            roof = dict({"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "kernel": kernel,
                         "algorithmic_bytes_per_env": per_env, "algorithmic_bytes_per_launch": per_env * N, "num_envs": N,
                         "launch_sampling": f"every {max(1, args.profile_stride)}th launch of a separate stamping loop after the timed batches",
                         "note": "dispatch-timestamp HIP events on the launch stream (a GPU spin in front of each stamped step keeps the queue "
                                 "from draining while the host pays for the stamped launch).  An event-stamped dispatch measures ~1.5 us "
                                 "(65 536 envs) / ~7 us (1 M envs) longer than the same kernel in the rocprofv3 --kernel-trace --stats "
                                 "summary of this command (profiles/): it completes with a system-scope release that back-to-back launches "
                                 "do not pay, so achieved / frac are lower bounds.  traffic: PMC FETCH_SIZE / WRITE_SIZE of this "
                                 "kernel from two child runs of this file under rocprofv3 --pmc (traffic_source), null when skipped"}, **m)

    if rank == 0:
        out = {
            "metric": "env-steps/sec", "value": global_envs * args.steps / batch, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": batch / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "go2_12dof_full_manager_stack" if go2 else args.config, "num_envs_per_gpu": N, "global_num_envs": global_envs,
                       "dofs": D, "reward_terms": T, "termination_terms": len(env.managers["termination"].term_cfg), "command_managers": len(env.managers["command"]),
                       "obs_width": int(env.observation_space.shape[0]), "scene": "synthetic (gf_synth_scene_step)", "parallelism": f"env-shard x{world}",
                       "stats_allreduce_every_steps": (args.reduce_every if dist_on else None), "dist_backend": (dist_backend if dist_on else None),
                       # dmabuf IPC for RCCL across processes on this driver: launch.spawn_ranks sets 0 unless the environment says otherwise
                       "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                       # SURVEY.md 8d's input spec would reset 2-8 % of the envs per step; the stand-in physics is set to about 0.2 % (episodes of a
                       # few hundred steps: the step, not the reset path, is what is timed) - the measured fraction of the last timed step:
                       "resets_last_step_frac": reset_frac,
                       "setup_steps_before_warmup": PRIMING_STEPS, "fused_post_physics": fused, "observation_output": OBS_OUTPUT,
                       "launches_per_step": env._trace.n_ops if env._trace is not None else None},
            "timing": {"batches": len(times), "timed_s": sum(times), "batch_ms_median": batch * 1e3, "batch_ms_min": min(times) * 1e3,
                       "batch_ms_max": max(times) * 1e3, "rule": f"batches of exactly {args.steps} steps (barrier + synchronize both sides, max over ranks) "
                                                                  f"until >= {MIN_TIMED_S} s are timed; median batch reported"},
            "roofline": roof,
            "rccl_ranks": (dist.get_world_size() if dist_on else 1),   # read back from the process group, not from --gpus
            "ranks": per_rank,
        }
    del env
    if world == 1 and not rehearsal:
        if not args.no_sweep and go2:
            # BASELINE.json quotes the metric "@ 4096–65536 envs": the other end of the range and the middle, same workload, same
            # timing rule (informational; `value` is the 65 536-env figure unless --num-envs says otherwise)
            out["sweep"] = []
            for n_s in (4096, 16384):
                if n_s == N:
                    continue
                env_s = make_env(n_s)
                env_s.seed(1234)
                env_s.reset()
                acts_s = [torch.randn(n_s, 12, generator=g).to(gs.device) for _ in range(8)]
                for i in range(PRIMING_STEPS + args.warmup):
                    env_s.step(acts_s[i % 8])
                ts = timed_batches(env_s, acts_s, args.steps)
                b = statistics.median(ts)
                out["sweep"].append({"num_envs": n_s, "value": n_s * args.steps / b, "unit": "env-steps/s", "ms_per_step": b / args.steps * 1e3, "batches": len(ts)})
                del env_s
        if not args.no_hbm_point and not args.no_profile and go2 and N != HBM_POINT_ENVS:
            # the HBM-honest point: at 65 536 envs a step's 53 MB working set stays in the 256 MiB Infinity Cache between kernels;
            # at 1 048 576 envs (0.85 GB) every launch streams from HBM
            n_h = HBM_POINT_ENVS
            env_h = make_env(n_h)
            env_h.seed(1234)
            env_h.reset()
            acts_h = [torch.randn(n_h, 12, generator=g).to(gs.device) for _ in range(4)]
            for i in range(PRIMING_STEPS + 10):
                env_h.step(acts_h[i % 4])
            ts = timed_batches(env_h, acts_h, min(args.steps, 100))
            b = statistics.median(ts) / min(args.steps, 100)
            pe = post_bytes_per_env(12, T, 3, 48, 1)
            m = stamp_pass(env_h, acts_h, pe * n_h)
            if m is not None:
                out["roofline_hbm"] = dict({"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "kernel": kernel, "num_envs": n_h,
                                            "algorithmic_bytes_per_env": pe, "algorithmic_bytes_per_launch": pe * n_h,
                                            "ms_per_step": b * 1e3, "value": n_h / b,
                                            "note": "same run, same kernel, 1 048 576 envs: the working set no longer fits the Infinity Cache"}, **m)
            del env_h
        if not args.no_pmc and not args.no_profile and go2 and out.get("roofline") is not None:
            # last of the GPU legs (the child processes need the device to themselves): PMC traffic of the dominant kernel
            sync()
            traffic, how = pmc_traffic(N)
            out["roofline"]["traffic"] = traffic
            out["roofline"]["traffic_source"] = how
            if traffic:
                out["roofline"]["traffic_over_algorithmic"] = traffic / out["roofline"]["algorithmic_bytes_per_launch"]
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(N, min(16, os.cpu_count() or 1))   # the GPU box gives one GPU's share of the host: 16 cores
            except Exception as ex:  # the baseline is a reported side figure: never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 1, "kind": "port", "sample": f"failed: {ex!r}"}
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    os.close(result_fd)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
