#!/usr/bin/env python3
"""development: where the host time of the ORDINARY (phase by phase, unrecorded) step goes — cProfile over 1 000 steps of the headline
config with recording switched off.   python tools/host_profile_ordinary.py [num_envs]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
os.environ["GF_NO_TRACE"] = "1"
import torch
from genesis_forge_amd import gs, tasks

gs.set_device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = tasks.bench_env(n)
env.build(); env.seed(1); env.reset()
acts = [torch.randn(n, 12, device="cuda") for _ in range(4)]
for i in range(100):
    env.step(acts[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1000):
    env.step(acts[i % 4])
torch.cuda.synchronize()
print(f"ordinary step, {n} envs: {(time.perf_counter() - t0) / 1000 * 1e6:.1f} us/step (recorded: {env._trace is not None})")
pr = cProfile.Profile()
pr.enable()
for i in range(1000):
    env.step(acts[i % 4])
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
