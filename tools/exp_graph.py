#!/usr/bin/env python3
"""Step time of the Go2 command config at 4 096 / 16 384 / 65 536 envs, best of 5 x 1000 steps: plain launches (default) or, with
GF_GRAPH=1, the recorded step as one hipGraphLaunch.  The third row of profiles/r02_b_hipgraph_static_experiment.jsonl
(GF_GRAPH_STATIC=1: the instantiated graph replayed with FROZEN arguments, no SetParams, no host-side packing — the best case a
device-resident step counter could reach) needed a one-line diagnostic change in gf_run_ops_graph that is not in the product:
    if (getenv("GF_GRAPH_STATIC") && g && g->exec) return (int)hipGraphLaunch(g->exec, stream);
Without it the variable is ignored."""
import os, sys, time, json
sys.path.insert(0, '/root/repo/genesis-forge_amd')
import torch
from genesis_forge_amd import gs, tasks
from genesis_forge_amd.managers import ObservationManager
ObservationManager.default_output = "static"
gs.set_device("cuda:0")
for n in (4096, 16384, 65536):
    env = tasks.bench_env(n); env.build(); env.seed(1234); env.reset()
    g = torch.Generator().manual_seed(0)
    acts = [torch.randn(n, 12, generator=g).to(gs.device) for _ in range(8)]
    for i in range(40): env.step(acts[i % 8])
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(1000): env.step(acts[i % 8])
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 1000)
    # host-only cost: time enqueue without sync
    print(json.dumps({"n": n, "graph": os.environ.get("GF_GRAPH"), "static": os.environ.get("GF_GRAPH_STATIC"), "us_per_step": best * 1e6}), flush=True)
