#!/usr/bin/env python3
"""Host-time profile (cProfile) of the gait task WITH the example's reset() override — recorded up to the reset, the user's reset() by
index list, native tail segments.      python tools/host_profile_override.py [num_envs]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
import torch
from genesis_forge_amd import gs, tasks

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
gs.set_device("cuda:0")
env = tasks.Go2GaitTrainingCurriculumEnv(num_envs=n, scene_kwargs=dict(ang_noise=0.05, seed=1234, contact_prob=0.001, contact_force=40.0))
env.build(); env.seed(1); env.reset()
acts = [torch.randn(n, 12, device=gs.device) for _ in range(4)]
for i in range(60):
    env.step(acts[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1000):
    env.step(acts[i % 4])
torch.cuda.synchronize()
print(f"n={n}: {(time.perf_counter() - t0) / 1000 * 1e6:.1f} us/step; recorded={env._trace is not None}, tail segments={sorted((env._trace.tail_seg or {}).keys()) if env._trace else None}")
pr = cProfile.Profile()
pr.enable()
for i in range(1000):
    env.step(acts[i % 4])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
