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
