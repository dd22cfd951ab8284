#!/usr/bin/env python3
"""Long-run check of the recorded step: N steps of the bench workload (and of the gait task with window output), device memory and host
RSS sampled along the way — a leak in the per-step machinery (fresh-tensor pool, statistics ring, log fillers) shows as growth.
    python tools/soak_steps.py [steps]"""
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
import torch
from genesis_forge_amd import gs, tasks
from genesis_forge_amd.managers import ObservationManager

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
gs.set_device("cuda:0")
for name, n, mode in (("go2_cmd_65536", 65536, "fresh"), ("gait", 8192, "window"), ("gait", 8192, "fresh")):
    ObservationManager.default_output = mode
    env = tasks.BASELINE_CONFIGS[name][1](n)
    env.build(); env.seed(1); env.reset()
    d = env.action_space.shape[0]
    acts = [torch.randn(n, d, device="cuda") for _ in range(4)]
    samples = []
    t0 = time.perf_counter()
    for i in range(steps):
        obs, rew, te, tr, ex = env.step(acts[i & 3])
        if i % 1000 == 0:
            _ = ex["episode"].get("Rewards / action_rate")   # a training loop reads its logs now and then
        if i in (steps // 10, steps // 2, steps - 1):
            torch.cuda.synchronize()
            samples.append((i, torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10))
    dt = time.perf_counter() - t0
    print(f"{name} n={n} output={mode}: {steps} steps, {dt / steps * 1e6:.1f} us/step; (step, MiB allocated, MiB reserved, host MiB): {samples}", flush=True)
    del env
