#!/usr/bin/env python3
"""Where the HOST time of a recorded step goes (cProfile over the replay loop).   python tools/host_profile.py <config> <num_envs>"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
import torch
from genesis_forge_amd import gs, tasks
from genesis_forge_amd.managers import ObservationManager

ObservationManager.default_output = os.environ.get("GF_OBS_OUTPUT", "fresh")
cfg = sys.argv[1] if len(sys.argv) > 1 else "go2_cmd"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
gs.set_device("cuda:0")
env = tasks.BASELINE_CONFIGS[cfg][1](n)
env.build()
if os.environ.get("GF_DIST_FORCE") == "1":   # the process-group path with one rank (RCCL), reduce_every from GF_REDUCE_EVERY (32)
    import torch.distributed as dist
    from genesis_forge_amd import distributed as gfd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29632")
    dist.init_process_group("nccl", rank=0, world_size=1)
    gfd.attach(env, reduce_every=int(os.environ.get("GF_REDUCE_EVERY", "32")), force=True, lockstep_reads=True)
env.seed(1); env.reset()
d = env.action_space.shape[0]
acts = [torch.randn(n, d, device=gs.device) for _ in range(8)]
for i in range(50):
    env.step(acts[i % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2000):
    env.step(acts[i % 8])
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{cfg} n={n}: enqueue {t_enq / 2000 * 1e6:.1f} us/step (host), with drain {t_all / 2000 * 1e6:.1f} us/step")
pr = cProfile.Profile()
pr.enable()
for i in range(2000):
    env.step(acts[i % 8])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("GF_PROFILE_ROWS", "14")))
