#!/usr/bin/env python3
"""Which kernel the fused post-physics launch of a BASELINE config uses, and its structure signature in the notation of
csrc/gf_post_programs.h (paste it there to register a static program).     python tools/describe_config.py gait [num_envs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
import torch
from genesis_forge_amd import gs, tasks

name = sys.argv[1] if len(sys.argv) > 1 else "go2_cmd"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
gs.set_device("cuda:0")
env = tasks.BASELINE_CONFIGS[name][1](n)
env.build()
env.reset()
d = env.action_space.shape[0]
for _ in range(6):
    env.step(torch.zeros(n, d, device=gs.device))
tr = env._trace
print("recorded:", tr is not None, "ops:", tr.n_ops if tr else None, "fused:", bool(tr and tr.post_refs is not None))
if tr is not None and tr.post_refs is not None:
    print(env.backend.post_describe(tr.post_refs))
