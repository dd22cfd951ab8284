#!/usr/bin/env python3
"""
bench.py — env-steps/sec of the ManagedEnvironment manager-step pipeline on MI355X.

A "step" is one full ManagedEnvironment.step() (action → synthetic scene tick → termination → reward →
command → masked reset → observation) over one batch of envs resident in HBM: BASELINE.json's
Go2 12-DOF config with the full Reward/Termination/Command manager stack (6 reward terms, 2 termination
terms, velocity command, 48-wide observation), synthetic data.  Weak scaling: every rank owns
``--num-envs`` envs; the only cross-rank traffic is the per-step logging all-reduce (RCCL).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  ``roofline`` is for the dominant kernel — the fused post-physics launch that
contains the reward fold (566 algorithmic B/env for this config; 268 B/env if the step runs unfused and
the stand-alone reward kernel is the one measured) — bytes ÷ its average duration measured with HIP events on the launch
stream during the timed region.  ``cpu_baseline`` times the CPU oracle (oracle/, a scalar C port of the
reference algorithm) on the host cores of the same box, on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


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


def make_env(num_envs: int):
    from envs import Go2CommandDirectionEnv

    env = Go2CommandDirectionEnv(num_envs=num_envs, max_episode_length_s=20, scene_kwargs=dict(ang_noise=0.05, seed=1234))
    env.build()
    return env


def cpu_baseline(num_envs: int, budget_s: float = 12.0) -> dict:
    """Time the oracle (kind "port": scalar C restatement of the reference managers, 1 thread) on this box's host."""
    import torch
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend

    old_dev, old_backend = gs.device, nat._backend
    gs.set_device("cpu")
    nat.set_backend(OracleBackend(os.path.join(ROOT, "oracle", "libgf_oracle.so")))
    try:
        env = make_env(num_envs)
        env.reset()
        g = torch.Generator().manual_seed(0)
        acts = [torch.randn(num_envs, 12, generator=g) for _ in range(4)]
        for i in range(2):
            env.step(acts[i % 4])
        t0 = time.perf_counter()
        steps = 0
        while True:
            env.step(acts[steps % 4])
            steps += 1
            if time.perf_counter() - t0 > budget_s or steps >= 2000:
                break
        dt = time.perf_counter() - t0
    finally:
        nat.set_backend(old_backend)
        gs.device = old_dev
    return {"value": num_envs * steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{steps} steps x {num_envs} envs of the same Go2 full-stack workload, oracle/libgf_oracle.so single thread, "
                      f"{os.cpu_count()} host cores present"}


def cpu_baseline_multicore(num_envs: int, workers: int, budget_s: float = 8.0) -> dict | None:
    """The same port on `workers` host cores: the path shards by env, so each worker process steps its own shard of
    num_envs / workers envs with the single-threaded oracle (no GPU in the workers); the figure is the sum."""
    import subprocess

    shard = max(1, num_envs // workers)
    # GF_DEVICE=cpu: the workers never call torch.cuda.is_available() (which opens the GPU; the box allows few processes on it)
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", GF_DEVICE="cpu", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="",
               CUDA_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(shard), str(budget_s)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(workers)]
    total, ok = 0.0, 0
    for p in procs:
        try:
            out, _ = p.communicate(timeout=budget_s * 4 + 120)
            total += float(out.strip().splitlines()[-1])
            ok += 1
        except Exception:
            p.kill()
    if ok != workers:
        return None
    return {"value": total, "unit": "env-steps/s", "cores": workers,
            "sample": f"{workers} worker processes x {shard} envs each, ~{budget_s:.0f} s, oracle single-threaded per worker"}


def cpu_worker(shard: int, budget_s: float) -> None:
    import torch

    torch.set_num_threads(1)
    print(cpu_baseline(shard, budget_s)["value"], flush=True)


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(int(sys.argv[2]), float(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--num-envs", type=int, default=65536, help="envs per GPU (weak scaling)")
    ap.add_argument("--reduce-every", type=int, default=32, help="recorded steps per logging all-reduce (world > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the 4096 / 16384-env side measurements")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event stamping of the dominant kernel")
    ap.add_argument("--profile-stride", type=int, default=0,
                    help="stamp every k-th launch of the dominant kernel in the timed region; 0 (default) = steps // 10, at least 8: "
                         "a stamped launch drains the queue and costs ≈ 35 µs of a ≈ 20 µs step, so ten samples per run keep the "
                         "throughput being measured within a few per cent of an unstamped run (1 = every launch)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import distributed as gfd
    from genesis_forge_amd import gs

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the manager phases only exist as HIP kernels")
    # one rank per GPU over RCCL ("nccl" on ROCm).  GF_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer
    # GPUs than ranks (ranks then share devices; RCCL refuses that)
    dist_backend = os.environ.get("GF_DIST_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    os.environ["LOCAL_RANK"] = str(local) if dist_backend != "nccl" else os.environ.get("LOCAL_RANK", "0")
    rank, world = gfd.init_from_env(dist_backend)
    gs.set_device(f"cuda:{local}")
    backend = nat.get_backend()

    N = args.num_envs
    env = make_env(N)
    # multi-GPU: the statistics rows of 32 steps share one all-reduce (fewer, larger collectives; bench reads its log on every
    # rank at the same step, which is what reduce_every > 1 asks for — see distributed.attach)
    gfd.attach(env, global_num_envs=N * world, reduce_every=args.reduce_every)
    env.seed(1234 + rank)
    env.reset()
    g = torch.Generator().manual_seed(1234 + rank)
    acts = [torch.randn(N, 12, generator=g).to(gs.device) for _ in range(8)]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, not warm-up: an env records its step over its first two steps and the first replay loads the fused kernel's code
    # object — done here so that even `--warmup 0` times the steady state (the W warm-up steps below are run on top, as asked)
    PRIMING_STEPS = 8
    for i in range(PRIMING_STEPS):
        env.step(acts[i % 8])
    for i in range(args.warmup):
        env.step(acts[i % 8])
    barrier()
    fused = env._trace is not None and env._trace.post_refs is not None
    prof_phase = nat.GF_PHASE_POST if fused else nat.GF_PHASE_REWARD
    if args.profile_stride <= 0:
        args.profile_stride = max(8, args.steps // 10)
    if not args.no_profile:
        backend.set_option(nat.GF_OPT_PROFILE_STRIDE, max(1, args.profile_stride))
        backend.profile_begin(prof_phase, args.steps)
    t0 = time.perf_counter()
    for i in range(args.steps):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    prof_ms, prof_n = (0.0, 0)
    if not args.no_profile:
        prof_ms, prof_n = backend.profile_end()
    if world > 1:
        t = torch.tensor([elapsed], device=gs.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # keep the logging path honest: read one step's global statistics
    _ = dict(env.extras["episode"])

    if rank == 0:
        rm = env.managers["reward"]
        T = sum(1 for c in rm.cfg.values() if c.weight != 0)
        per_env = post_bytes_per_env(12, T, 3, 48, 1) if fused else reward_bytes_per_env(12, T, 3)
        bytes_per_launch = per_env * N
        kernel = "gf::reward_kernel<3>"
        if fused:
            # which kernel gf_post_physics_step launches for this config: "program <id> (<name>): <signature>"
            what = backend.post_describe(env._trace.post_refs).split(":")[0]
            kernel = f"gf::post_ws_kernel, {what} (termination+reward+command+reset+observe fused)"
        roof = None
        traffic, traffic_src = None, None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if fused and os.path.exists(pmc_file):
            # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE), collected in
            # separate rocprofv3 --pmc passes over this same command and committed under profiles/ (r01_pmc_traffic.md)
            rec = json.load(open(pmc_file)).get(str(N))
            if rec and ("Prog" in rec["kernel"]) == ("program 0" not in what):  # counters were taken on this same kernel
                traffic, traffic_src = rec["traffic_bytes"], "profiles/r01_pmc_traffic.md"
        # the same kernel's average in the committed rocprofv3 --kernel-trace --stats summary of this command (unperturbed by
        # the event stamping: stamped launches start on a drained queue and run ≈ 2 µs longer)
        rocprof_avg_us, rocprof_src = None, None
        import csv
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_bench_N{N}_*kernel_stats.csv")))[::-1]:
            for row in csv.DictReader(open(path)):
                if ("post_ws_kernel" in row["Name"] or "post_kernel" in row["Name"]) == fused and \
                        (("post_" in row["Name"]) if fused else ("reward_kernel" in row["Name"])):
                    rocprof_avg_us, rocprof_src = float(row["AverageNs"]) / 1e3, os.path.relpath(path, ROOT)
                    break
            if rocprof_avg_us is not None:
                break
        if prof_n > 0:
            avg_s = prof_ms / prof_n / 1e3
            achieved = bytes_per_launch / avg_s / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel, "algorithmic_bytes_per_env": per_env, "avg_launch_us": avg_s * 1e6, "launches": prof_n, "launch_sampling": f"every {max(1, args.profile_stride)}th launch of the timed region",
                    "note": "stamped launches start on a drained queue and run ~2 us longer than the unstamped ones (DESIGN.md 4.3): "
                            "achieved / frac are lower bounds, rocprof_avg_launch_us is the unperturbed average",
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "rocprof_avg_launch_us": rocprof_avg_us, "rocprof_source": rocprof_src}
        out = {
            "metric": "env-steps/sec", "value": world * N * args.steps / elapsed, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "go2_12dof_full_manager_stack", "num_envs_per_gpu": N, "global_num_envs": N * world, "dofs": 12,
                       "reward_terms": T, "termination_terms": 2, "command_managers": 1, "obs_width": 48,
                       "scene": "synthetic (gf_synth_scene_step)", "parallelism": f"env-shard x{world}",
                       "stats_allreduce_every_steps": (args.reduce_every if world > 1 else None),
                       "setup_steps_before_warmup": PRIMING_STEPS},
            "roofline": roof,
        }
        if world == 1 and not args.no_sweep:
            # BASELINE.json quotes the metric "@ 4096–65536 envs": the other end of the range (and the middle), same workload, measured
            # after the timed region above, without launch stamps (informational; `value` is the 65 536-env figure unless --num-envs)
            out["sweep"] = []
            for n_s in (4096, 16384):
                if n_s == N:
                    continue
                env_s = make_env(n_s)
                env_s.seed(1234)
                env_s.reset()
                acts_s = [torch.randn(n_s, 12, generator=g).to(gs.device) for _ in range(4)]
                for i in range(args.warmup):
                    env_s.step(acts_s[i % 4])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    env_s.step(acts_s[i % 4])
                torch.cuda.synchronize()
                dt_s = time.perf_counter() - t0
                out["sweep"].append({"num_envs": n_s, "value": n_s * args.steps / dt_s, "unit": "env-steps/s", "ms_per_step": dt_s / args.steps * 1e3})
                del env_s
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N)
            workers = min(16, os.cpu_count() or 1)   # the GPU box gives one GPU's share of the host: 16 cores
            if workers > 1:
                multi = cpu_baseline_multicore(N, workers)
                if multi is not None:
                    out["cpu_baseline"]["multicore"] = multi
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
