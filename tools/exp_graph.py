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
