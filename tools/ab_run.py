#!/usr/bin/env python3
"""development: step time of configs under variants of the library (tools/ab_build.sh), interleaved on one box.
    python tools/ab_run.py <config,config,...> <variant> [<variant> ...]   ("product" = the in-tree library)
Each (variant, config) runs in a child process; rounds alternate over the variants so box drift hits all alike."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(variant, config, steps):
    os.environ["GF_JIT"] = "off"
    sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from genesis_forge_amd import _native, gs, tasks
    if variant != "product":
        lib = os.path.join(ROOT, "tools", "_ab", variant, "libgf_step.so")
        _native.lib_path = lambda: lib
    gs.set_device("cuda:0")
    if os.environ.get("GF_OBS_OUTPUT"):   # "static" / "ring": another output contract than the default
        from genesis_forge_amd.managers import ObservationManager
        ObservationManager.default_output = os.environ["GF_OBS_OUTPUT"]
    name, _, size = config.partition("@")   # "go2_cmd@1048576": the config at another size
    if name == "gait_override":   # the gait task with the example's reset() override (tools/bench_reset_override.py)
        n = int(size) if size else 8192
        env = tasks.Go2GaitTrainingCurriculumEnv(num_envs=n, scene_kwargs=dict(ang_noise=0.05, seed=1234, contact_prob=0.001, contact_force=40.0))
    else:
        n, factory = tasks.BASELINE_CONFIGS[name]
        n = int(size) if size else n
        env = factory(n)
    env.build(); env.seed(1); env.reset()
    d = env.action_space.shape[0]
    acts = [torch.randn(n, d, device="cuda") for _ in range(4)]
    for i in range(50):
        env.step(acts[i % 4])
    best = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            env.step(acts[i % 4])
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / steps * 1e6)
    print(json.dumps({"variant": variant, "config": config, "us_min": min(best), "us_med": sorted(best)[2]}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    else:
        configs = sys.argv[1].split(",")
        variants = sys.argv[2:]
        for rnd in range(2):
            for c in configs:
                for v in variants:
                    subprocess.run([sys.executable, __file__, "--child", v, c, "300"], check=True)
