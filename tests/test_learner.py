"""SURVEY.md §8f-5, first slice: the RL-library side of a step.

* RolloutStorage: the step's own kernels write `observations[t+1]`, `rewards[t]`, `dones[t]` (time-major rows) — parity is
  against what rsl_rl's storage does, a torch `copy_` of the step's returned tensors into the same layout.
* GradientAllReduce: one flat-bucket all-reduce averages the policy gradients of env-sharded ranks (world_size 2 over gloo
  here, RCCL on GPUs): equal to the gradient of the unsharded batch."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(kind, n):
    from genesis_forge_amd import tasks

    if kind == "go2":
        return tasks.Go2CommandDirectionEnv(num_envs=n, max_episode_length_s=0.4, cmd_resample_s=0.2, scene_kwargs=dict(ang_noise=0.3, seed=3))
    if kind == "go2_hist":
        return tasks.Go2CommandDirectionEnv(num_envs=n, max_episode_length_s=0.4, cmd_resample_s=0.2, history=3, contacts=True, obs_noise=True,
                                            scene_kwargs=dict(ang_noise=0.3, seed=3))
    if kind in ("go2_user_reward", "go2_user_obs"):   # a Python-level term in the step: termination launch → callable → fused launch (rows
        env = tasks.Go2CommandDirectionEnv(num_envs=n, max_episode_length_s=0.4, cmd_resample_s=0.2, history=2, contacts=True,   # included); an
                                           scene_kwargs=dict(ang_noise=0.3, seed=3))                                          # observation callable: chains
        base = env.config

        def config():
            base()
            from genesis_forge_amd.managers import ObservationManager, RewardManager
            if kind == "go2_user_reward":
                rc = {k: {"weight": v.weight, "fn": v.fn, "params": dict(v.params)} for k, v in env.reward_manager.cfg.items()}
                rc["user_height"] = {"weight": 0.3, "fn": lambda e: torch.tanh(e.robot.get_pos()[:, 2])}
                env.managers["reward"] = None
                env.reward_manager = RewardManager(env, logging_enabled=True, cfg=rc)
            else:
                om = env.observation_manager
                oc = {k: {"fn": v.fn, "params": dict(v.params), "scale": v.scale, "noise": v.noise} for k, v in om.cfg.items()}
                oc["user_xy"] = {"fn": lambda env: env.robot.get_pos()[:, :2] * 2.0}
                env.managers["observation"].remove(om)
                env.observation_manager = ObservationManager(env, cfg=oc, history_len=2)

        env.config = config
        return env
    if kind == "gait":
        return tasks.Go2GaitTrainingEnv(num_envs=n, max_episode_length_s=0.4, scene_kwargs=dict(ang_noise=0.3, seed=3, contact_prob=0.05))
    if kind == "gait_curriculum":   # reset() override: recorded up to the reset, the tail (and the rollout write) runs phase by phase
        return tasks.Go2GaitTrainingCurriculumEnv(num_envs=n, max_episode_length_s=0.4, scene_kwargs=dict(ang_noise=0.3, seed=3, contact_prob=0.05))
    raise KeyError(kind)


def _check_rollout(dev, kind, n, trace, horizon=5, steps=17, fuse=True, output="fresh"):
    """``output``: "fresh" keeps a history of H > 1 frames as a ring and gathers it (the gather stores the storage row as its second
    destination when the step is fused), "static" shifts it inside the observation launch (the fused launch stores the row)."""
    from genesis_forge_amd.learner import RolloutStorage
    from genesis_forge_amd.managers import ObservationManager

    old, ObservationManager.default_output = ObservationManager.default_output, output
    try:   # (the managers are created by env.config(), i.e. inside build())
        env = _make(kind, n)
        env.trace_enabled = trace
        env.fuse_post_physics = fuse
        env.build()
    finally:
        ObservationManager.default_output = old
    assert all(m._output == output for m in env.managers["observation"])
    env.seed(7)
    obs, _ = env.reset()
    store = RolloutStorage(env, horizon).attach()
    store.begin(obs)
    W = env.observation_space.shape[0]
    # the reference-side storage: plain torch copies of what step() returns (rsl_rl RolloutStorage.add_transitions)
    ref_obs = torch.zeros(horizon + 1, n, W, device=dev)
    ref_rew = torch.zeros(horizon, n, device=dev)
    ref_done = torch.zeros(horizon, n, dtype=torch.bool, device=dev)
    ref_obs[0].copy_(obs)
    g = torch.Generator().manual_seed(1)
    d = env.action_space.shape[0]
    dones = 0
    for k in range(steps):
        t = k % horizon
        if t == 0 and k > 0:
            ref_obs[0].copy_(ref_obs[horizon])
        obs, rew, term, trunc, _ = env.step(torch.randn(n, d, generator=g).to(dev))
        ref_obs[t + 1].copy_(obs)
        ref_rew[t].copy_(rew)
        ref_done[t].copy_(term | trunc)
        dones += int((term | trunc).sum())
        assert store.step == t + 1 and store.full == (t + 1 == horizon)
        for name, a, b in (("observations", store.observations[: t + 2], ref_obs[: t + 2]), ("rewards", store.rewards[: t + 1], ref_rew[: t + 1]),
                           ("dones", store.dones[: t + 1], ref_done[: t + 1])):
            assert torch.equal(a, b), f"{name} differ from the torch copy_ storage at step {k}"
    assert dones > 0
    return env


@pytest.mark.parametrize("kind,trace,output", [("go2", False, "fresh"), ("go2", True, "fresh"), ("go2_hist", True, "fresh"), ("gait", True, "fresh"),
                                               ("gait_curriculum", True, "fresh"), ("go2_hist", True, "static"), ("go2_hist", False, "static"),
                                               ("gait", True, "static"), ("go2_user_reward", True, "fresh"), ("go2_user_obs", True, "fresh"),
                                               ("go2_user_reward", True, "static")])
def test_rollout_storage_rows_cpu(oracle_backend, kind, trace, output):
    env = _check_rollout("cpu", kind, 70, trace, output=output)
    assert (env._trace is not None) == trace
    if kind in ("go2_hist", "gait"):
        assert all(m._unrolled == (output == "fresh") for m in env.managers["observation"] if m._history_len > 1)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,trace,fuse,output", [("go2", False, True, "fresh"), ("go2", True, True, "fresh"), ("go2", True, False, "fresh"),
                                                    ("go2_hist", True, True, "fresh"), ("gait", True, True, "fresh"), ("gait", True, False, "fresh"),
                                                    ("gait_curriculum", True, True, "fresh"), ("go2_hist", True, True, "static"),
                                                    ("gait", True, True, "static"), ("gait", True, False, "static"),
                                                    ("go2_user_reward", True, True, "fresh"), ("go2_user_reward", True, True, "static"),
                                                    ("go2_user_obs", True, True, "fresh")])
def test_rollout_storage_rows_hip(hip_backend, kind, trace, fuse, output):
    env = _check_rollout("cuda", kind, 1000, trace, fuse=fuse, output=output)
    tr = env._trace
    assert (tr is not None) == trace
    if trace and kind != "gait_curriculum":
        fused = fuse and kind != "go2_user_obs"   # (a rollout row out of a manager that observes behind the fused launch: phase chains)
        assert (tr.post_refs is not None) == fused
        if fused:
            assert tr.post_refs.rollout, "the fused post-physics launch stores the rollout rows itself"
            assert bool(tr.post_refs.flags & 1) == (kind == "go2_user_reward"), "a Python-level reward term: termination as a launch of its own"


def test_rollout_storage_refuses_a_window_mode_observation(oracle_backend):
    """output="window" hands out a strided view of the history buffer; the storage's rows are copied by the step's launch from contiguous
    rows — it says so instead of copying the wrong floats, also when the mode is switched on later."""
    from envs import Go2CommandDirectionEnv
    from genesis_forge_amd.learner import RolloutStorage

    env = Go2CommandDirectionEnv(num_envs=20, history=3, scene_kwargs=dict(seed=2))
    env.build()
    env.observation_manager.output = "window"
    with pytest.raises(ValueError, match="window"):
        RolloutStorage(env, 4)
    env.observation_manager.output = "fresh"
    st = RolloutStorage(env, 4).attach()
    st.begin(env.reset()[0])
    for _ in range(3):
        env.step(torch.zeros(20, 12))
    env.observation_manager.output = "window"
    with pytest.raises(ValueError, match="strided"):
        for _ in range(3):
            env.step(torch.zeros(20, 12))


def test_rollout_write_abi_validation(oracle_backend):
    import ctypes as C
    from genesis_forge_amd import _native as nat

    a = nat.GfRolloutArgs()
    a.num_envs, a.obs_width = 4, 3
    out = torch.zeros(4, 3)
    a.obs_out = out.data_ptr()
    with pytest.raises(nat.GfError):
        oracle_backend.call("rollout_write", a)     # obs_out without obs
    a.obs = torch.ones(4, 3).data_ptr()
    src = torch.arange(12, dtype=torch.float32).reshape(4, 3)
    a.obs = src.data_ptr()
    oracle_backend.call("rollout_write", a)
    assert torch.equal(out, src)


# ---- gradient all-reduce ------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _grad_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
    import torch.distributed as dist
    from genesis_forge_amd.learner import ActorCriticMLP, GradientAllReduce

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), GF_DEVICE="cpu")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)            # replicas start DIFFERENT on purpose: broadcast_parameters must fix that
    net = ActorCriticMLP(48, 12)
    sync = GradientAllReduce(net.parameters())
    sync.broadcast_parameters(0)
    if world == 1:
        torch.manual_seed(100)
        net = ActorCriticMLP(48, 12)
        sync = GradientAllReduce(net.parameters())
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(5)
    N = 64
    for it in range(3):
        obs = torch.randn(N, 48, generator=g)
        tgt_a, tgt_v = torch.randn(N, 12, generator=g), torch.randn(N, 1, generator=g)
        lo, hi = (rank * N // world, (rank + 1) * N // world)
        sync.zero_grad()
        loss = ((net.act_mean(obs[lo:hi]) - tgt_a[lo:hi]) ** 2).mean() + ((net.evaluate(obs[lo:hi]) - tgt_v[lo:hi]) ** 2).mean() + (net.std ** 2).sum()
        loss.backward()
        work = sync.average(async_op=world > 1)
        sync.wait()
        grads = sync.bucket.clone()
        opt.step()
    torch.save({"grads": grads, "params": torch.cat([p.detach().reshape(-1) for p in net.parameters()]), "nbytes": sync.nbytes},
               os.path.join(out_dir, f"rank{rank}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_gradient_allreduce_matches_unsharded_batch():
    ctx = mp.get_context("spawn")
    with tempfile.TemporaryDirectory() as d1, tempfile.TemporaryDirectory() as d2:
        p = ctx.Process(target=_grad_worker, args=(0, 1, _free_port(), d1))
        p.start(); p.join(200)
        assert p.exitcode == 0
        port = _free_port()
        procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, d2)) for r in range(2)]
        for q in procs:
            q.start()
        for q in procs:
            q.join(200)
            assert q.exitcode == 0
        full = torch.load(os.path.join(d1, "rank0.pt"))
        shards = [torch.load(os.path.join(d2, f"rank{r}.pt")) for r in range(2)]
    assert 1.4e6 < full["nbytes"] < 1.7e6, "the 512-256-128 actor + critic of the reference configs is about 1.5 MB of gradients"
    assert torch.equal(shards[0]["grads"], shards[1]["grads"]) and torch.equal(shards[0]["params"], shards[1]["params"]), "replicas diverged"
    # mean over two equal shards == mean over the whole batch (up to f32 summation order)
    assert torch.allclose(shards[0]["grads"], full["grads"], atol=1e-6, rtol=1e-4)
    assert torch.allclose(shards[0]["params"], full["params"], atol=1e-5, rtol=1e-4)


# ---------------------------------------------------------------------------------------------------------------------------------
# §8f-5 remainder: the policy's rows of a transition, the time-out bootstrap, GAE / returns, the gradient bucket on a GPU
# ---------------------------------------------------------------------------------------------------------------------------------
def _torch_compute_returns(rewards, values, dones, last_values, gamma, lam, normalize=True):
    """rsl_rl RolloutStorage.compute_returns, statement for statement (the torch reference of gf_gae)."""
    T = rewards.shape[0]
    returns = torch.zeros_like(rewards)
    advantage = 0
    for step in reversed(range(T)):
        next_values = last_values if step == T - 1 else values[step + 1]
        next_is_not_terminal = 1.0 - dones[step].float()
        delta = rewards[step] + next_is_not_terminal * gamma * next_values - values[step]
        advantage = delta + next_is_not_terminal * gamma * lam * advantage
        returns[step] = advantage + values[step]
    advantages = returns - values
    raw = advantages.clone()
    if normalize:
        advantages = (advantages - advantages.mean()) / (advantages.std() + 1e-8)
    return returns, advantages, raw


def _policy_rollout(dev, n, horizon=24, rollouts=2, trace=True):
    """A PPO-style collection loop on the Go2 task: act → env.step (the step's kernel writes obs / reward / done rows) → add_policy
    (policy rows + time-out bootstrap) … → compute_returns; next to it rsl_rl's storage semantics in plain torch."""
    from genesis_forge_amd.learner import ActorCriticMLP, RolloutStorage

    env = _make("go2", n)
    env.trace_enabled = trace
    env.build()
    env.seed(7)
    obs, _ = env.reset()
    ro = RolloutStorage(env, horizon).attach()
    ro.begin(obs)
    torch.manual_seed(3)
    net = ActorCriticMLP(48, 12).to(dev)
    g = torch.Generator().manual_seed(11)
    gamma, lam = 0.99, 0.95
    results = []
    for r in range(rollouts):
        ref = {k: [] for k in ("actions", "values", "logp", "mu", "sigma", "rewards", "dones")}
        for t in range(horizon):
            with torch.no_grad():
                mu = net.act_mean(obs)
                sigma = net.std.expand_as(mu).contiguous()
                actions = mu + sigma * torch.randn(mu.shape, generator=g).to(dev)
                values = net.evaluate(obs)
                logp = torch.distributions.Normal(mu, sigma).log_prob(actions).sum(dim=-1)
            obs, rew, term, trunc, extras = env.step(actions)
            ro.add_policy(actions, values, logp, mu, sigma, time_outs=trunc, gamma=gamma)
            # rsl_rl: PPO.process_env_step (bootstrap on time-outs) then RolloutStorage.add_transitions (copy_ of everything)
            rew_ref = rew.clone() + gamma * torch.squeeze(values * trunc.unsqueeze(1).float(), 1)
            for k, v in (("actions", actions), ("values", values.reshape(-1)), ("logp", logp), ("mu", mu), ("sigma", sigma), ("rewards", rew_ref),
                         ("dones", term | trunc)):
                ref[k].append(v.clone())
        with torch.no_grad():
            last_values = net.evaluate(obs)
        ro.compute_returns(last_values, gamma=gamma, lam=lam, normalize=True)
        ref = {k: torch.stack(v) for k, v in ref.items()}
        want_ret, want_adv, _raw = _torch_compute_returns(ref["rewards"], ref["values"], ref["dones"], last_values.reshape(-1), gamma, lam)
        results.append((ref, want_ret, want_adv, {k: getattr(ro, k).clone() for k in ("actions", "values", "actions_log_prob", "mu", "sigma", "rewards",
                                                                                      "dones", "returns", "advantages")}))
    return results, env


def _check_policy_rollout(results):
    for ref, want_ret, want_adv, got in results:
        for a, b in (("actions", "actions"), ("values", "values"), ("logp", "actions_log_prob"), ("mu", "mu"), ("sigma", "sigma"), ("dones", "dones")):
            assert torch.equal(ref[a], got[b]), f"{b} rows differ from the copy_ storage"
        assert torch.equal(ref["rewards"], got["rewards"]), "rewards after the time-out bootstrap differ"
        assert float((ref["rewards"].abs()).sum()) > 0
        assert torch.equal(want_ret, got["returns"]), f"returns differ from the torch loop: {(want_ret - got['returns']).abs().max()}"
        assert torch.allclose(want_adv, got["advantages"], atol=1e-5, rtol=1e-5), f"normalised advantages: {(want_adv - got['advantages']).abs().max()}"
    assert any(bool(r[0]["dones"].any()) for r in results), "no env finished an episode in any rollout: the done / time-out paths went untested"


def test_policy_rows_and_gae_cpu(oracle_backend):
    results, env = _policy_rollout("cpu", 70, horizon=12, rollouts=2)
    _check_policy_rollout(results)
    assert env._trace is not None


@pytest.mark.gpu
@pytest.mark.parametrize("n,horizon", [(70, 13), (1000, 24), (4096, 24)])
def test_policy_rows_and_gae_hip(hip_backend, n, horizon):
    results, env = _policy_rollout("cuda", n, horizon=horizon, rollouts=2)
    _check_policy_rollout(results)
    assert env._trace is not None and env._trace.post_refs is not None


def test_gae_abi_validation_and_shapes(oracle_backend):
    """gf_gae / gf_rollout_policy_write through the raw ABI (the oracle twin): refusals, T = 1, unnormalised advantages == returns - values."""
    import ctypes as C
    from genesis_forge_amd import _native as nat

    g = nat.GfGaeArgs()
    with pytest.raises(nat.GfError, match="GF_E_NULL"):
        oracle_backend.call("gae", g)
    N, T = 37, 1
    gen = torch.Generator().manual_seed(0)
    rew, val, last = torch.randn(T, N, generator=gen), torch.randn(T, N, generator=gen), torch.randn(N, generator=gen)
    dones = torch.rand(T, N, generator=gen) < 0.3
    ret, adv, mom = torch.zeros(T, N), torch.zeros(T, N), torch.zeros(2, dtype=torch.float64)
    g.num_envs, g.num_steps, g.gamma, g.lam = N, T, 0.9, 0.8
    g.rewards, g.values, g.dones, g.last_values = rew.data_ptr(), val.data_ptr(), dones.data_ptr(), last.data_ptr()
    g.returns, g.advantages = ret.data_ptr(), adv.data_ptr()
    g.normalize = 1
    with pytest.raises(nat.GfError, match="GF_E_NULL"):   # normalisation needs the moments scratch
        oracle_backend.call("gae", g)
    g.normalize, g.moments = 0, mom.data_ptr()
    oracle_backend.call("gae", g)
    want_ret, _adv, raw = _torch_compute_returns(rew, val, dones, last, 0.9, 0.8, normalize=False)
    assert torch.equal(ret, want_ret) and torch.equal(adv, raw)
    assert abs(float(mom[0]) - float(raw.double().sum())) < 1e-9
    g.num_steps = 0
    with pytest.raises(nat.GfError, match="GF_E_RANGE"):
        oracle_backend.call("gae", g)
    p = nat.GfRolloutPolicyArgs()
    p.num_envs, p.num_actions = N, 3
    out = torch.zeros(N, 3)
    p.actions_out = out.data_ptr()
    with pytest.raises(nat.GfError, match="GF_E_NULL"):   # a destination without its source
        oracle_backend.call("rollout_policy_write", p)


@pytest.mark.gpu
def test_gae_kernel_hip_equals_torch_at_size(hip_backend):
    """gf_gae alone at the benchmark size (65 536 envs x 24 steps): returns bit-identical to the torch loop, advantages to 1e-5."""
    from genesis_forge_amd import _native as nat

    N, T = 65536 + 37, 24
    gen = torch.Generator().manual_seed(1)
    rew, val, last = (torch.randn(T, N, generator=gen).cuda(), torch.randn(T, N, generator=gen).cuda(), torch.randn(N, generator=gen).cuda())
    dones = (torch.rand(T, N, generator=gen) < 0.05).cuda()
    ret, adv, mom = torch.zeros(T, N, device="cuda"), torch.zeros(T, N, device="cuda"), torch.zeros(2, dtype=torch.float64, device="cuda")
    g = nat.GfGaeArgs()
    g.num_envs, g.num_steps, g.gamma, g.lam, g.normalize = N, T, 0.99, 0.95, 1
    g.rewards, g.values, g.dones, g.last_values = rew.data_ptr(), val.data_ptr(), dones.data_ptr(), last.data_ptr()
    g.returns, g.advantages, g.moments = ret.data_ptr(), adv.data_ptr(), mom.data_ptr()
    hip_backend.call("gae", g)
    want_ret, want_adv, _raw = _torch_compute_returns(rew, val, dones, last, 0.99, 0.95)
    assert torch.equal(ret, want_ret)
    assert torch.allclose(adv, want_adv, atol=1e-5, rtol=1e-5)


@pytest.mark.gpu
def test_gradient_allreduce_on_rccl_group_of_one(hip_backend):
    """The gradient bucket on a GPU through real RCCL: a forced group of one rank runs every call of the path (in-place division,
    all-reduce of the flat 1.5 MB bucket — synchronous and asynchronous + wait —, the parameter broadcast); the sum over one rank
    is the identity, so gradients and the optimizer step equal a run without a process group."""
    import torch.distributed as dist
    from genesis_forge_amd.learner import ActorCriticMLP, GradientAllReduce

    def run(force):
        torch.manual_seed(100)
        net = ActorCriticMLP(48, 12).cuda()
        sync = GradientAllReduce(net.parameters(), force=force)
        sync.broadcast_parameters(0)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        g = torch.Generator().manual_seed(5)
        grads = []
        for it in range(3):
            obs, tgt = torch.randn(256, 48, generator=g).cuda(), torch.randn(256, 12, generator=g).cuda()
            sync.zero_grad()
            loss = ((net.act_mean(obs) - tgt) ** 2).mean() + (net.evaluate(obs) ** 2).mean() + (net.std ** 2).sum()
            loss.backward()
            work = sync.average(async_op=(it % 2 == 1))
            if it % 2 == 1:
                assert (work is not None) == force   # (a synchronous all_reduce returns no work object)
            sync.wait()
            grads.append(sync.bucket.clone())
            opt.step()
        return grads, torch.cat([p.detach().reshape(-1) for p in net.parameters()]), sync.nbytes

    want = run(False)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        got = run(True)
    finally:
        dist.destroy_process_group()
    assert 1.4e6 < got[2] < 1.7e6
    for a, b in zip(want[0], got[0]):
        assert torch.equal(a, b)
    assert torch.equal(want[1], got[1])
