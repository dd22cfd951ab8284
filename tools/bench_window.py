#!/usr/bin/env python3
"""The gait task's step with the four observation output contracts (fresh / static / ring / window) and what a policy's first layer pays
for reading the window's strided view instead of a contiguous tensor.   python tools/bench_window.py [num_envs]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
import torch
from genesis_forge_amd import gs, tasks
from genesis_forge_amd.managers import ObservationManager

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
gs.set_device("cuda:0")
os.environ.setdefault("GF_JIT", "off")
for mode in ("fresh", "static", "ring", "window"):
    ObservationManager.default_output = mode
    env = tasks.BASELINE_CONFIGS["gait"][1](n)
    env.build(); env.seed(1); env.reset()
    d = env.action_space.shape[0]
    acts = [torch.randn(n, d, device="cuda") for _ in range(4)]
    for i in range(40):
        obs = env.step(acts[i % 4])[0]
    best = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(200):
            obs = env.step(acts[i % 4])[0]
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / 200 * 1e6)
    line = {"config": "gait", "num_envs": n, "output": mode, "us_per_step_min": round(min(best), 1), "us_per_step_med": round(sorted(best)[2], 1),
            "obs_shape": list(obs.shape), "obs_stride": list(obs.stride()), "fused": env._trace is not None and env._trace.post_refs is not None}
    if mode in ("fresh", "window"):   # the policy's first layer (rsl_rl's 512-wide actor) on this step's observation
        w = torch.randn(512, obs.shape[1], device="cuda") * 0.01
        for _ in range(5):
            torch.nn.functional.linear(obs, w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            torch.nn.functional.linear(obs, w)
        torch.cuda.synchronize()
        line["first_layer_us"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
    print(json.dumps(line), flush=True)
    del env
