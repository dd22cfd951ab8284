"""
The RL-library side of a step (first slice): what rsl_rl's OnPolicyRunner does with a step's outputs and
with the policy gradients (call site examples/simple/train.py:125-129; policy config :37-79).  rsl_rl itself is a third-party
dependency and stays one; this module provides the two pieces of it that touch the hot path's data and the xGMI links:

* :class:`RolloutStorage` — time-major ``observations [T+1, N, W]``, ``rewards [T, N]``, ``dones [T, N]``.  rsl_rl fills its
  storage with three ``copy_`` launches per step; attached to a ``ManagedEnvironment`` the step's own kernel stores the three
  rows (``gf_rollout_write`` phase by phase, the fused post-physics launch writes them from the tile it holds) — no extra
  launch, no re-read of the observation.
* :class:`GradientAllReduce` — the multi-GPU half: every rank owns a shard of envs and a replica of the policy; after
  ``backward()`` the gradients of all parameters are averaged with ONE all-reduce over a flat bucket (RCCL over xGMI on GPUs;
  the 512-256-128 actor + critic MLPs of the reference configs are 1.5 MB), overlapped with nothing because nothing follows
  it but the optimizer step.  This is the first place of the pipeline where xGMI bandwidth rather than latency matters.
* :class:`ActorCriticMLP` — the policy of ``examples/*/train.py`` (ELU, hidden dims 512-256-128, learnable action std) as a
  plain torch module, so the two pieces above can be exercised without rsl_rl.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence

import torch

from . import _native as nat
from . import gs


class RolloutStorage:
    """Rollout rows of ``num_steps`` transitions for ``env.num_envs`` envs (layout of rsl_rl's RolloutStorage, time-major).

    ``observations[t]`` is the policy input of transition ``t`` (row 0: the observation the rollout starts from; row
    ``num_steps``: the bootstrap observation), ``rewards[t]`` / ``dones[t]`` its outcome.  ``attach()`` makes every
    ``env.step()`` write transition ``step``'s rows and advance; after ``num_steps`` steps ``full`` is true and the next step
    starts the next rollout (its row 0 is the previous rollout's row ``num_steps``)."""

    def __init__(self, env, num_steps: int, obs_name: str = "policy"):
        self.env, self.num_steps, self.obs_name = env, int(num_steps), obs_name
        n = env.num_envs
        om = next((m for m in env.managers["observation"] if m.name == obs_name), None)
        if om is None:
            raise ValueError(f"no ObservationManager named '{obs_name}'")
        self._om = om
        self.obs_width = int(om.observation_space.shape[0])
        self.observations = torch.zeros((self.num_steps + 1, n, self.obs_width), device=gs.device, dtype=torch.float32)
        self.rewards = torch.zeros((self.num_steps, n), device=gs.device, dtype=torch.float32)
        self.dones = torch.zeros((self.num_steps, n), device=gs.device, dtype=torch.bool)
        self.step = 0            # transitions written in the current rollout
        self._args = nat.GfRolloutArgs()
        self._args.num_envs, self._args.obs_width = n, self.obs_width

    @property
    def full(self) -> bool:
        return self.step >= self.num_steps

    def attach(self) -> "RolloutStorage":
        self.env._rollout = self
        self.env.invalidate_trace()
        return self

    def detach(self) -> None:
        if getattr(self.env, "_rollout", None) is self:
            self.env._rollout = None
            self.env.invalidate_trace()

    def begin(self, obs: torch.Tensor) -> None:
        """Start a rollout from ``obs`` (what ``env.reset()`` returned)."""
        self.observations[0].copy_(obs)
        self.step = 0

    def _next_rows(self, a: nat.GfRolloutArgs) -> None:
        """Point the descriptor at transition ``step``'s rows and advance (wrapping into the next rollout)."""
        if self.step >= self.num_steps:
            self.observations[0].copy_(self.observations[self.num_steps])
            self.step = 0
        t = self.step
        a.obs_out = self.observations.data_ptr() + (t + 1) * self.observations.stride(0) * 4
        a.reward_out = self.rewards.data_ptr() + t * self.rewards.stride(0) * 4
        a.done_out = self.dones.data_ptr() + t * self.dones.stride(0)
        self.step = t + 1

    def write(self, obs: torch.Tensor, reward: torch.Tensor, terminated: torch.Tensor, truncated: torch.Tensor) -> None:
        """One transition through ``gf_rollout_write`` (the phase-by-phase path; a recorded step fuses it, _trace.py)."""
        a = self._args
        a.obs, a.reward = obs.data_ptr(), reward.data_ptr()
        a.terminated, a.truncated = terminated.data_ptr(), truncated.data_ptr()
        self._next_rows(a)
        self._keep = (obs, reward, terminated, truncated)
        self.env.backend.call("rollout_write", a, owner=self)

    def _trace_patch(self, args, via_unroll=None):
        """Recorded step: advance the rows.  ``via_unroll`` = the policy ObservationManager when it keeps its history as a ring and
        the step is fused: the fused launch only holds the new frame, so the observation row is written by the manager's gather
        (second destination of gf_history_unroll) and the fused launch gets no observation row."""
        def patch(_actions, a=args, self=self, om=via_unroll):
            self._next_rows(a)
            if om is not None:
                om._unroll_args.out2, a.obs_out = a.obs_out, None

        return patch

    def _trace_native(self, args, pol, fused: bool) -> list:
        """`obs` follows the tensor the observation launch (or the gather of a ring-kept history) writes this step — that
        manager's rotation comes earlier in the table."""
        P = nat.GfReplayPatch
        if getattr(pol, "_unrolled", False):
            return [] if fused else [P(nat.GF_PATCH_COPY, 0, nat.field_addr(args, "obs"), None, nat.field_addr(pol._unroll_args, "out"))]
        return [P(nat.GF_PATCH_COPY, 0, nat.field_addr(args, "obs"), None, nat.field_addr(pol._args, "obs"))]


class ActorCriticMLP(torch.nn.Module):
    """rsl_rl's ``ActorCritic`` as configured by the reference's training scripts (examples/simple/train.py:56-62)."""

    def __init__(self, num_obs: int, num_actions: int, actor_hidden_dims: Sequence[int] = (512, 256, 128),
                 critic_hidden_dims: Sequence[int] = (512, 256, 128), init_noise_std: float = 1.0):
        super().__init__()

        def mlp(sizes):
            layers = []
            for i in range(len(sizes) - 1):
                layers.append(torch.nn.Linear(sizes[i], sizes[i + 1]))
                if i < len(sizes) - 2:
                    layers.append(torch.nn.ELU())
            return torch.nn.Sequential(*layers)

        self.actor = mlp([num_obs, *actor_hidden_dims, num_actions])
        self.critic = mlp([num_obs, *critic_hidden_dims, 1])
        self.std = torch.nn.Parameter(init_noise_std * torch.ones(num_actions))

    def act_mean(self, obs: torch.Tensor) -> torch.Tensor:
        return self.actor(obs)

    def evaluate(self, obs: torch.Tensor) -> torch.Tensor:
        return self.critic(obs)


class GradientAllReduce:
    """Average the gradients of ``params`` over the ranks of ``group`` with one collective.

    All gradients live in ONE flat, persistent bucket (``param.grad`` are views into it), so the step after ``backward()`` is a
    single ``all_reduce`` of ``numel * 4`` bytes — RCCL's ring over xGMI moves it at link speed instead of paying the launch
    and latency cost of one collective per tensor (the reference policies have 16 parameter tensors).  ``average()`` divides
    by the world size in the same pass.  With one rank (or no process group) it is a no-op, so a training script is the same
    on one GPU and on eight."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        import torch.distributed as dist

        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        dev, dt = self.params[0].device, self.params[0].dtype
        self.bucket = torch.zeros(sum(p.numel() for p in self.params), device=dev, dtype=dt)
        off = 0
        for p in self.params:   # gradients are views of the bucket from now on: backward() accumulates straight into it
            p.grad = self.bucket[off:off + p.numel()].view_as(p)
            off += p.numel()
        self._work = None

    @property
    def nbytes(self) -> int:
        return self.bucket.numel() * self.bucket.element_size()

    def zero_grad(self) -> None:
        self.bucket.zero_()      # one memset; optimizer.zero_grad(set_to_none=True) would drop the views

    def average(self, async_op: bool = False):
        """Sum over ranks, divide by the world size.  ``async_op``: returns at once; call ``wait()`` before the optimizer step."""
        if self.world == 1:
            return None
        import torch.distributed as dist

        for p in self.params:    # a parameter whose grad was replaced (set_to_none) would silently leave the bucket
            if p.grad is None or p.grad.data_ptr() < self.bucket.data_ptr() or p.grad.data_ptr() >= self.bucket.data_ptr() + self.nbytes:
                raise RuntimeError("a parameter's .grad no longer lives in the bucket: use GradientAllReduce.zero_grad(), not set_to_none")
        self.bucket.div_(self.world)
        self._work = dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return self._work

    def wait(self) -> None:
        if self._work is not None:
            self._work.wait()
            self._work = None

    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every replica start from rank ``src``'s weights."""
        if self.world == 1:
            return
        import torch.distributed as dist

        for p in self.params:
            dist.broadcast(p.data, src=src, group=self.group)
