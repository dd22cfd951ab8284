#!/usr/bin/env python3
"""env-steps/s of every BASELINE.json config (genesis_forge_amd/tasks.py restatements of the reference's examples) on one GPU.

bench.py measures the headline config only (its JSON contract); this tool gives the per-config table of DESIGN.md:
    python tools/bench_configs.py [--steps 200] [--configs simple,go2_cmd,...] [--num-envs N]
One JSON line per config: N, µs/step, env-steps/s, whether the step was recorded, whether its post-physics phases run as the
fused launch, and the number of native ops one recorded step replays.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))

def _configs():
    from genesis_forge_amd.tasks import BASELINE_CONFIGS   # (name -> (BASELINE.json size, factory)), shared with the parity tests

    return BASELINE_CONFIGS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--configs", default="")
    ap.add_argument("--num-envs", type=int, default=0, help="override every config's size")
    args = ap.parse_args()
    import torch
    from genesis_forge_amd import gs
    from genesis_forge_amd.managers import ObservationManager

    ObservationManager.default_output = os.environ.get("GF_OBS_OUTPUT", "fresh")   # as bench.py: the default output contract

    if not torch.cuda.is_available():
        raise SystemExit("needs a ROCm GPU")
    gs.set_device("cuda:0")
    cfgs = _configs()
    names = [c for c in args.configs.split(",") if c] or list(cfgs)
    for name in names:
        n0, make = cfgs[name]
        n = args.num_envs or n0
        env = make(n)
        env.build()
        env.seed(1234)
        env.reset()
        d = env.action_space.shape[0]
        g = torch.Generator().manual_seed(0)
        acts = [torch.randn(n, d, generator=g).to(gs.device) for _ in range(8)]
        for i in range(args.warmup):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        log = dict(env.extras["episode"])
        tr = env._trace
        print(json.dumps({"config": name, "num_envs": n, "us_per_step": dt / args.steps * 1e6, "env_steps_per_s": n * args.steps / dt,
                          "recorded": tr is not None, "fused_post": bool(tr is not None and tr.post_refs is not None),
                          "ops_per_step": tr.n_ops if tr is not None else None,
                          "resets_last_step_frac": sum(float(v) for k, v in log.items() if k.startswith("Terminations /"))}), flush=True)
        del env


if __name__ == "__main__":
    main()
