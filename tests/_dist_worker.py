"""Worker for test_distributed_gloo.py: one rank of an env-sharded run on CPU (gloo) with the oracle backend."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genesis-forge_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_shard(rank, world, port, out_dir, n_global, steps, sizes, reduce_every=1, read_lag=0, mutate_at=None, user_term=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import distributed as gfd
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend
    from envs import Go2CommandDirectionEnv

    gs.set_device("cpu")
    nat.set_backend(OracleBackend(os.path.join(ROOT, "oracle", "libgf_oracle.so")))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    start = sum(sizes[:rank])
    count = sizes[rank]
    env = Go2CommandDirectionEnv(num_envs=count, max_episode_length_s=1, cmd_resample_s=0.3, contacts=True, history=2, obs_noise=True,
                                 scene_kwargs=dict(ang_noise=0.3, seed=3))
    if user_term:  # a Python-level reward term: the recorded step is cut around the call (per-env state only, so sharding commutes)
        from genesis_forge_amd.managers import RewardManager
        base_config = env.config

        def config():
            base_config()
            rc = {k: {"weight": v.weight, "fn": v.fn, "params": dict(v.params)} for k, v in env.reward_manager.cfg.items()}
            rc["user_height"] = {"weight": 0.3, "fn": lambda e: torch.tanh(e.robot.get_pos()[:, 2])}
            env.managers["reward"] = None
            env.reward_manager = RewardManager(env, logging_enabled=True, cfg=rc)

        env.config = config
    env.build()
    env.seed(5)
    gfd.attach(env, reduce_every=reduce_every, lockstep_reads=True)   # (every rank reads the same steps' logs)
    assert env.env_offset == start and env.global_num_envs == n_global
    env.reset()
    g = torch.Generator().manual_seed(0)
    outs, held = [], []
    for t in range(steps):
        if mutate_at is not None and t == mutate_at:   # a curriculum step: every rank takes it at the same step
            env.reward_manager.cfg["action_rate"].weight = -0.5
            env.velocity_command.range["lin_vel_x"][1] = 3.0
        act = torch.randn(n_global, 12, generator=g)[start:start + count].contiguous()
        o, r, te, tr, ex = env.step(act)
        held.append((o.clone(), r.clone(), te.clone(), tr.clone(), ex["episode"]))
        # logs are read `read_lag` steps late (a training loop reads them at the end of an iteration), on every rank alike
        while held and (len(held) > read_lag or t == steps - 1):
            h = held.pop(0)
            outs.append(h[:4] + ({k: float(v) for k, v in h[4].items()},))
    torch.save({"start": start, "count": count, "outs": outs, "traced": env._trace is not None,
                "cuts": len(env._trace.splits) if env._trace is not None else 0}, os.path.join(out_dir, f"rank{rank}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_probe_shard(rank, world, port, out_dir, n_global, steps, sizes, reduce_every, probes):
    """Nobody reads a log except at the ``probes`` steps, where every rank reads that step's log and the reward manager's
    last-episode means (a curriculum's read).  Resets are rare in this config, so some probes find their answer in a ring row,
    others in the row that was carried into ``last_reset`` on the device when its batch was recycled."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import distributed as gfd
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend
    from envs import Go2CommandDirectionEnv

    gs.set_device("cpu")
    nat.set_backend(OracleBackend(os.path.join(ROOT, "oracle", "libgf_oracle.so")))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = sum(sizes[:rank]), sizes[rank]
    env = Go2CommandDirectionEnv(num_envs=count, max_episode_length_s=2, cmd_resample_s=0.3, contacts=False, history=2, obs_noise=True,
                                 scene_kwargs=dict(ang_noise=0.0, seed=3))
    env.build()
    env.seed(5)
    gfd.attach(env, reduce_every=reduce_every, lockstep_reads=True)
    env.reset()
    g = torch.Generator().manual_seed(0)
    out, resets = {}, []
    for t in range(steps):
        act = torch.randn(n_global, 12, generator=g)[start:start + count].contiguous()
        o, r, te, tr, ex = env.step(act)
        resets.append(int((te | tr).sum()))
        if t in probes:
            rm = env.reward_manager
            out[t] = ({k: float(v) for k, v in ex["episode"].items()}, {name: rm.last_episode_mean_reward(name) for name in rm.cfg})
    torch.save({"probes": out, "resets": resets, "traced": env._trace is not None}, os.path.join(out_dir, f"rank{rank}.pt"))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_rank0_reader(rank, world, port, out_dir, reduce_every):
    """A rank-0-only logger with K > 1: rank 0 reads a fresh step's log, rank 1 never does.  Without ``lockstep_reads`` the read must
    RAISE on rank 0 (closing the open batch is a collective the other rank never enters); the run then goes on."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from genesis_forge_amd import _native as nat
    from genesis_forge_amd import distributed as gfd
    from genesis_forge_amd import gs
    from oracle_backend import OracleBackend
    from envs import Go2CommandDirectionEnv

    gs.set_device("cpu")
    nat.set_backend(OracleBackend(os.path.join(ROOT, "oracle", "libgf_oracle.so")))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env = Go2CommandDirectionEnv(num_envs=33 + 4 * rank, max_episode_length_s=1, cmd_resample_s=0.3, scene_kwargs=dict(ang_noise=0.3, seed=3))
    env.build()
    env.seed(5)
    gfd.attach(env, reduce_every=reduce_every)
    env.reset()
    g = torch.Generator().manual_seed(rank)
    raised, old_ok = False, False
    logs = []
    for t in range(3 * reduce_every + 5):
        _o, _r, _te, _tr, ex = env.step(torch.randn(env.num_envs, 12, generator=g))
        logs.append(ex["episode"])
        if rank == 0 and t == reduce_every + 2 and env._trace is not None:
            try:
                dict(ex["episode"])
            except RuntimeError as e:
                raised = "collective" in str(e)
        if rank == 0 and t == 3 * reduce_every + 4:
            dict(logs[t - 2 * reduce_every - 2])   # a step whose batch closed long ago: rank-local, no collective
            old_ok = True
    torch.save({"raised": raised, "old_ok": old_ok, "traced": env._trace is not None}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
