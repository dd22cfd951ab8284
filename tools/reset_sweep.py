#!/usr/bin/env python3
"""Step time of bench.py's workload against the fraction of envs that reset per step (SURVEY.md §8d's input spec resets 2-8 % per step,
the timed configs about 0.2 %): the stand-in physics' attitude noise is raised until more envs fall over.
    python tools/reset_sweep.py [num_envs]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
import torch
from genesis_forge_amd import gs
from genesis_forge_amd.tasks import Go2CommandDirectionEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
gs.set_device("cuda:0")
for noise in (0.05, 0.2, 0.4, 0.7, 1.2):
    env = Go2CommandDirectionEnv(num_envs=n, max_episode_length_s=20, scene_kwargs=dict(ang_noise=noise, seed=1234))
    env.build(); env.seed(1); env.reset()
    acts = [torch.randn(n, 12, device="cuda") for _ in range(4)]
    for i in range(60):
        env.step(acts[i % 4])
    resets = 0
    best = []
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(300):
            env.step(acts[i % 4])
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / 300 * 1e6)
    fr = []
    for i in range(20):
        _o, _r, te, tr, _e = env.step(acts[i % 4])
        fr.append(float((te | tr).float().mean()))
    print(json.dumps({"num_envs": n, "ang_noise": noise, "resets_per_step_frac": round(sum(fr) / len(fr), 4), "us_per_step_min": round(min(best), 2),
                      "us_per_step_med": round(sorted(best)[1], 2), "recorded": env._trace is not None}), flush=True)
    del env
