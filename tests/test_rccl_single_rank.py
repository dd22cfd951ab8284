"""The collective path of the logging statistics on REAL RCCL, with the only group a one-GPU box can form: one rank.

The multi-rank path is covered over gloo on the CPU (tests/test_distributed_gloo.py); what gloo cannot show is whether the calls
this package makes are ones RCCL accepts — float64 all-reduce of rows of the device vector ring, asynchronous work objects waited
on the launch stream, batched rows (reduce_every > 1), the barrier / MAX all-reduce bench.py uses.  A group of one runs every one
of them (the sum over one rank is the identity), so the logs must equal a run without a process group."""
import os
import socket

import pytest
import torch


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(reduce_every, steps=70):
    from genesis_forge_amd import distributed as gfd
    from genesis_forge_amd import tasks

    env = tasks.Go2CommandDirectionEnv(num_envs=1000, max_episode_length_s=0.4, cmd_resample_s=0.2, contacts=True, scene_kwargs=dict(ang_noise=0.3, seed=3))
    env.build()
    if reduce_every:
        gfd.attach(env, reduce_every=reduce_every, force=True, lockstep_reads=True)
        assert env.stats.group is not None
    env.seed(5)
    env.reset()
    g = torch.Generator().manual_seed(0)
    logs, held = [], []
    for t in range(steps):
        out = env.step(torch.randn(1000, 12, generator=g).to("cuda"))
        held.append(out[4]["episode"])
        if len(held) > 5:   # logs read a few steps late, as a training loop does
            logs.append({k: float(v) for k, v in held.pop(0).items()})
    logs += [{k: float(v) for k, v in h.items()} for h in held]
    assert env._trace is not None
    return logs


@pytest.mark.gpu
@pytest.mark.parametrize("reduce_every", [1, 8, 32])
def test_statistics_allreduce_on_rccl_group_of_one(hip_backend, reduce_every):
    import torch.distributed as dist

    want = _run(0)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        t = torch.tensor([1.5], device="cuda", dtype=torch.float64)   # what bench.py's timing does
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert float(t.item()) == 1.5
        got = _run(reduce_every)
    finally:
        dist.destroy_process_group()
        for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)
    assert len(got) == len(want)
    resets = 0
    for t, (a, b) in enumerate(zip(got, want)):
        assert a.keys() == b.keys(), f"log keys differ at step {t}"
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-9 + 1e-9 * abs(b[k]), (t, k, a[k], b[k])
        resets += sum(1 for k in a if k.startswith("Rewards /"))
    assert resets > 0
