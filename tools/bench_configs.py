#!/usr/bin/env python3
"""env-steps/s of every BASELINE.json config (genesis_forge_amd/tasks.py restatements of the reference's examples) on one GPU.

bench.py measures the headline config only (its JSON contract); this tool gives the per-config table of DESIGN.md:
    python tools/bench_configs.py [--steps 200] [--configs simple,go2_cmd,...] [--num-envs N]
One JSON line per config: N, µs/step, env-steps/s, whether the step was recorded, whether its post-physics phases run as the
fused launch, and the number of native ops one recorded step replays.
"""
import argparse
import json
import gc
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
    ap.add_argument("--scene", default="synthetic", choices=("synthetic", "genesis_like"),
                    help="genesis_like: the test double with Genesis' public surface only (tests/genesis_like.py: fresh getter tensors, "
                         "envs_idx setters, no gf_* extras); the line then also carries the double's own cost per tick")
    ap.add_argument("--no-trace", action="store_true", help="time the ordinary (phase by phase) step")
    args = ap.parse_args()
    os.environ.setdefault("GF_JIT", "sync")   # a config without a built-in program is timed on its compiled one from the first timed step on
    import torch
    from genesis_forge_amd import gs
    from genesis_forge_amd.managers import ObservationManager

    ObservationManager.default_output = os.environ.get("GF_OBS_OUTPUT", "fresh")   # as bench.py: the default output contract

    if not torch.cuda.is_available():
        raise SystemExit("needs a ROCm GPU")
    gs.set_device("cuda:0")
    cfgs = _configs()
    names = [c for c in args.configs.split(",") if c] or list(cfgs)
    scene_cls = None
    if args.scene == "genesis_like":
        sys.path.insert(0, os.path.join(ROOT, "tests"))   # the double is test infrastructure
        from genesis_like import GenesisLikeScene
        from genesis_forge_amd import tasks

        GenesisLikeScene.poison = False   # (the NaN-poisoning of stale getter tensors is a test aid, not part of a tick)
        scene_cls = GenesisLikeScene
    for name in names:
        n0, make = cfgs[name]
        n = args.num_envs or n0
        if scene_cls is not None:
            with tasks.use_scene(scene_cls):
                env = make(n)
        else:
            env = make(n)
        env.trace_enabled = not args.no_trace
        env.build()
        env.seed(1234)
        env.reset()
        d = env.action_space.shape[0]
        g = torch.Generator().manual_seed(0)
        acts = [torch.randn(n, d, generator=g).to(gs.device) for _ in range(8)]
        for i in range(args.warmup):
            env.step(acts[i % 8])
        gc.collect()   # (the previous config's env — gigabytes of tensors in reference cycles — is released here, not inside the timed loop)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        log = dict(env.extras["episode"])
        tr = env._trace
        info = getattr(env, "_program_info", None) or {}
        try:
            post_kernel = env.backend.post_describe(tr.post_refs).split(":")[0] if tr is not None and tr.post_refs is not None else None
        except Exception:
            post_kernel = None
        row = {"config": name, "scene": args.scene, "num_envs": n, "us_per_step": dt / args.steps * 1e6, "env_steps_per_s": n * args.steps / dt,
               "recorded": tr is not None, "fused_post": bool(tr is not None and tr.post_refs is not None),
               "ops_per_step": tr.n_ops if tr is not None else None,
               "resets_last_step_frac": sum(float(v) for k, v in log.items() if k.startswith("Terminations /")),
               "post_kernel": post_kernel, "jit_compile_s": info.get("compile_s"), "jit_error": info.get("error")}
        if scene_cls is not None:
            row["why_not_recorded"] = env._untraceable
            ad = env._adapter
            plan = tr.scene_plan if tr is not None else ad.plan()
            row["getters_per_tick"] = len(plan)
            # the L0 share: what a tick costs on the simulator's side of the boundary, with no manager work at all —
            # control_dofs_position + scene.step() + every getter of the plan once + the nonzero() its envs_idx setters force
            # (synchronising, as in the step) + those setters for the done envs (the masks of the last timed step, every tick).
            # On real Genesis each of these calls is a Taichi kernel plus a tensor conversion; here each is one torch op.
            am, tm = env.managers["action"], env.managers["termination"]
            m1, m2 = tm._terminated_buf.clone(), tm._truncated_buf.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                env.robot.control_dofs_position(am._actions, am.dofs_idx)
                env.scene.step()
                ad.refetch(plan)
                ad.push_done(m1, m2)
            torch.cuda.synchronize()
            row["l0_us_per_tick"] = (time.perf_counter() - t0) / args.steps * 1e6
            row["done_envs_in_l0_loop"] = int((m1 | m2).sum())
            row["manager_us_per_step"] = row["us_per_step"] - row["l0_us_per_tick"]
        print(json.dumps(row), flush=True)
        del env


if __name__ == "__main__":
    main()
