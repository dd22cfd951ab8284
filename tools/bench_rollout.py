#!/usr/bin/env python3
"""§8f-5: cost of filling the RL library's rollout storage — the step's own kernel stores the rows (learner.RolloutStorage)
versus three torch copy_ launches after the step (what rsl_rl's RolloutStorage.add_transitions does).
    python tools/bench_rollout.py [num_envs] [config]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
import torch
from genesis_forge_amd import gs, tasks
from genesis_forge_amd.learner import RolloutStorage
from genesis_forge_amd.managers import ObservationManager

ObservationManager.default_output = "static"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = sys.argv[2] if len(sys.argv) > 2 else "go2_cmd"
gs.set_device("cuda:0")
T = 24   # num_steps_per_env of examples/simple/train.py:75


def run(mode):
    env = tasks.BASELINE_CONFIGS[cfg][1](n)
    env.build()
    env.seed(1234)
    obs, _ = env.reset()
    d = env.action_space.shape[0]
    W = env.observation_space.shape[0]
    store = None
    if mode == "fused":
        store = RolloutStorage(env, T).attach()
        store.begin(obs)
    else:
        o = torch.zeros(T + 1, n, W, device=gs.device); r = torch.zeros(T, n, device=gs.device); dn = torch.zeros(T, n, dtype=torch.bool, device=gs.device)
    g = torch.Generator().manual_seed(0)
    acts = [torch.randn(n, d, generator=g).to(gs.device) for _ in range(8)]

    def step(i):
        ob, rw, te, tr, _ = env.step(acts[i % 8])
        if mode == "copies":
            t = i % T
            o[t + 1].copy_(ob); r[t].copy_(rw); dn[t].copy_(te | tr)

    for i in range(48):
        step(i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(480):
            step(i)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 480)
    tr = env._trace
    print(json.dumps({"config": cfg, "num_envs": n, "mode": mode, "us_per_step": best * 1e6, "ops_per_step": tr.n_ops if tr else None,
                      "rollout_in_fused_launch": bool(tr and tr.post_refs is not None and tr.post_refs.rollout)}), flush=True)


for mode in ("none", "copies", "fused"):
    run(mode)


def bench_returns():
    """The end of a rollout: rsl_rl's compute_returns as the torch loop it is (≈ 8 launches per step + the normalisation) versus
    gf_gae; and the policy's five rows of a transition as five copy_ launches versus gf_rollout_policy_write."""
    from genesis_forge_amd import _native as nat

    be = nat.get_backend()
    A = 12
    gen = torch.Generator().manual_seed(1)
    rew, val, last = (torch.randn(T, n, generator=gen).to(gs.device), torch.randn(T, n, generator=gen).to(gs.device), torch.randn(n, generator=gen).to(gs.device))
    dones = (torch.rand(T, n, generator=gen) < 0.05).to(gs.device)
    ret, adv, mom = torch.zeros(T, n, device=gs.device), torch.zeros(T, n, device=gs.device), torch.zeros(2, dtype=torch.float64, device=gs.device)
    g = nat.GfGaeArgs()
    g.num_envs, g.num_steps, g.gamma, g.lam, g.normalize = n, T, 0.99, 0.95, 1
    g.rewards, g.values, g.dones, g.last_values = rew.data_ptr(), val.data_ptr(), dones.data_ptr(), last.data_ptr()
    g.returns, g.advantages, g.moments = ret.data_ptr(), adv.data_ptr(), mom.data_ptr()

    def torch_loop():
        returns = torch.zeros_like(rew)
        advantage = 0
        for step in reversed(range(T)):
            nv = last if step == T - 1 else val[step + 1]
            nt = 1.0 - dones[step].float()
            delta = rew[step] + nt * 0.99 * nv - val[step]
            advantage = delta + nt * 0.99 * 0.95 * advantage
            returns[step] = advantage + val[step]
        a = returns - val
        return (a - a.mean()) / (a.std() + 1e-8)

    def timeit(f, reps=30):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    us_native = timeit(lambda: be.call("gae", g))
    us_torch = timeit(torch_loop)
    # the five policy rows
    src = [torch.randn(n, A, device=gs.device), torch.randn(n, device=gs.device), torch.randn(n, device=gs.device), torch.randn(n, A, device=gs.device),
           torch.rand(n, A, device=gs.device)]
    dst = [torch.zeros(T, n, A, device=gs.device), torch.zeros(T, n, device=gs.device), torch.zeros(T, n, device=gs.device), torch.zeros(T, n, A, device=gs.device),
           torch.zeros(T, n, A, device=gs.device)]
    tout = (torch.rand(n, device=gs.device) < 0.01)
    p = nat.GfRolloutPolicyArgs()
    p.num_envs, p.num_actions, p.gamma = n, A, 0.99
    p.actions, p.values, p.log_prob, p.mu, p.sigma = (x.data_ptr() for x in src)
    p.actions_out, p.values_out, p.log_prob_out, p.mu_out, p.sigma_out = (x[3].data_ptr() for x in dst)
    p.time_outs, p.reward_row = tout.data_ptr(), rew[3].data_ptr()

    def copies():
        rew[3].add_(0.99 * src[1] * tout.float())
        for s_, d_ in zip(src, dst):
            d_[3].copy_(s_)

    us_pol_native = timeit(lambda: be.call("rollout_policy_write", p), reps=200)
    us_pol_torch = timeit(copies, reps=200)
    bytes_gae = T * n * (9 + 8 + 8)
    print(json.dumps({"what": "returns", "num_envs": n, "T": T, "gf_gae_us": us_native, "torch_loop_us": us_torch,
                      "gae_algorithmic_MB": bytes_gae / 1e6, "gae_TBps": bytes_gae / us_native / 1e6,
                      "policy_rows_native_us": us_pol_native, "policy_rows_copies_us": us_pol_torch}), flush=True)


bench_returns()
