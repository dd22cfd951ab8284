"""
Multi-GPU: env sharding + the one collective of the path.

One process per GPU (``torch.distributed``, backend ``nccl`` == RCCL on ROCm, xGMI inside a node).
Every manager buffer is indexed by env and no term reads another env's data, so rank ``r`` of ``R``
simply owns a contiguous shard of envs with its own scene and RNG stream; rewards, observations and
done masks stay sharded and are consumed by the local policy replica.  The only cross-env values are
the logging / curriculum scalars (per-term episode means, termination fractions, reset counts): the
per-step statistics block is summed over ranks with a single all-reduce (~350 B, latency bound,
issued asynchronously behind the step's kernels), so every rank sees identical global values and
takes identical curriculum decisions.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def shard(global_num_envs: int, rank: int, world_size: int) -> tuple[int, int]:
    """(first env, env count) of ``rank``'s contiguous shard; the first ``global % world`` ranks get one extra."""
    base, rem = divmod(global_num_envs, world_size)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def init_from_env(backend: Optional[str] = None) -> tuple[int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def attach(env, group=None, global_num_envs: Optional[int] = None, reduce_every: int = 1, force: bool = False, lockstep_reads: bool = False) -> None:
    """Make ``env``'s logging statistics global: sums over the ranks of ``group`` (default group if None).
    ``env.num_envs`` stays the local shard size; ``env.global_num_envs`` is the denominator of fractions.

    ``reduce_every`` = K batches the statistics rows of K consecutive recorded steps into ONE all-reduce (K·392 B instead of
    K collectives of 392 B: the payload is latency bound either way, and enqueueing a collective costs the host about as much
    as the rest of a 25 µs step).  Per-step values are identical.  With K = 1 (default) log reads are rank-local and free of
    ordering rules; with K > 1 reading the log of a step whose batch is still open closes the batch early — a collective —
    so all ranks must then read the same steps (training loops that log on every rank, curricula — which run on every rank
    by construction — and ``env.stats.flush_reduce()`` are fine; a rank-0-only logger should keep K = 1 or read only
    steps older than K).  K must divide 64 and be at most 32.

    Collective discipline (what may NOT be rank-local once a group is attached).  With K = 1 every step — recorded or ordinary —
    issues exactly one all-reduce of one GF_STATS_VECTOR_LEN row, so ranks may drop and re-record their steps independently
    (a weight mutated on one rank only, a descriptor going dirty on one rank); the calls that add a collective of their own are
    ``env.reset(...)`` OUTSIDE a step (its log needs the global means) and ``attach`` itself — call those on every rank.
    With K > 1 the batches are tied to the recorded step: anything that invalidates it (``invalidate_trace``: a mutated weight /
    param / range, ``reset([ids])`` or ``resample_command`` between steps, a re-seed) closes the open batch with a collective
    and must therefore happen on every rank at the same step, as must reading the log of a step whose batch is still open.
    K > 1 is for lock-step training loops only; it stays opt-in until RCCL scaling has been measured on a real node.
    ``lockstep_reads`` (K > 1): the caller promises that every rank reads the same steps' logs, so a read may close an open batch by
    itself; without it such a read RAISES (a rank-0-only logger would otherwise wait for the other ranks forever)."""
    if reduce_every < 1 or reduce_every > 32 or 64 % reduce_every != 0:
        raise ValueError("reduce_every must divide 64 and be at most 32")
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        env.global_num_envs = env.num_envs if global_num_envs is None else global_num_envs
        return   # (``force``: take the collective path even in a group of one — how the RCCL calls are exercised on a one-GPU box)
    env.stats.group = group if group is not None else dist.group.WORLD
    env.stats.reduce_every = reduce_every
    env.stats.lockstep_reads = bool(lockstep_reads)
    if global_num_envs is None:
        n = torch.tensor([env.num_envs], dtype=torch.int64, device=env.stats.device)
        dist.all_reduce(n, group=group)
        global_num_envs = int(n.item())
    env.global_num_envs = global_num_envs
    # Philox is keyed by the GLOBAL env id: the sharded run reproduces the unsharded one env for env
    counts = torch.zeros(dist.get_world_size(group), dtype=torch.int64, device=env.stats.device)
    counts[dist.get_rank(group)] = env.num_envs
    dist.all_reduce(counts, group=group)
    offset = int(counts[: dist.get_rank(group)].sum().item())
    env.env_offset = offset
    if hasattr(env.scene, "env_offset"):
        env.scene.env_offset = offset
    env.invalidate_trace()
