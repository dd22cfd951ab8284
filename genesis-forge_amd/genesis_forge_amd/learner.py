"""
The RL-library side of a step (first slice): what rsl_rl's OnPolicyRunner does with a step's outputs and
with the policy gradients (call site examples/simple/train.py:125-129; policy config :37-79).  rsl_rl itself is a third-party
dependency and stays one; this module provides the two pieces of it that touch the hot path's data and the xGMI links:

* :class:`RolloutStorage` — time-major ``observations [T+1, N, W]``, ``rewards [T, N]``, ``dones [T, N]``.  rsl_rl fills its
  storage with three ``copy_`` launches per step; attached to a ``ManagedEnvironment`` the step's own kernel stores the three
  rows (``gf_rollout_write`` phase by phase, the fused post-physics launch writes them from the tile it holds) — no extra
  launch, no re-read of the observation.
  The policy's half of a transition (actions, value, log-probability, action mean / std — five more ``copy_`` launches in rsl_rl's
  ``add_transitions``, after ``rewards += gamma * values * time_outs``) is one ``gf_rollout_policy_write`` launch
  (``add_policy``), and the end-of-rollout return computation (rsl_rl ``compute_returns``: a backwards loop of eight
  elementwise launches per step, then the advantage normalisation) is ``gf_gae`` (``compute_returns``): one lane per env
  walks its T steps, rows are coalesced, the recurrence is the torch loop's arithmetic operation for operation.
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
        if getattr(om, "output", None) == "window" and om._history_len > 1:
            raise ValueError("RolloutStorage copies contiguous observation rows: an ObservationManager with output='window' hands out a "
                             "strided view — use output='fresh' / 'static' for the manager the storage follows")
        self.obs_width = int(om.observation_space.shape[0])
        self.observations = torch.zeros((self.num_steps + 1, n, self.obs_width), device=gs.device, dtype=torch.float32)
        self.rewards = torch.zeros((self.num_steps, n), device=gs.device, dtype=torch.float32)
        self.dones = torch.zeros((self.num_steps, n), device=gs.device, dtype=torch.bool)
        self.step = 0            # transitions written in the current rollout
        self._args = nat.GfRolloutArgs()
        self._args.num_envs, self._args.obs_width = n, self.obs_width
        # the policy's rows and the return computation (allocated on first use: a storage that only takes the env's rows stays small)
        self.num_actions = None
        self.actions = self.values = self.actions_log_prob = self.mu = self.sigma = self.returns = self.advantages = None
        self._pol_args = nat.GfRolloutPolicyArgs()
        self._gae_args = nat.GfGaeArgs()
        self._moments = None

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
        if not obs.is_contiguous():   # (an ObservationManager switched to output="window" after this storage was made)
            raise ValueError("RolloutStorage copies contiguous observation rows: the observation it follows is a strided view (output='window')")
        a = self._args
        a.obs, a.reward = obs.data_ptr(), reward.data_ptr()
        a.terminated, a.truncated = terminated.data_ptr(), truncated.data_ptr()
        self._next_rows(a)
        self._keep = (obs, reward, terminated, truncated)
        self.env.backend.call("rollout_write", a, owner=self)

    # -- the policy's half of a transition, returns ---------------------------------------------------------------------------
    def _ensure_policy_rows(self, num_actions: int) -> None:
        if self.actions is not None and self.num_actions == num_actions:
            return
        T, n = self.num_steps, self.env.num_envs
        z = lambda *shape: torch.zeros(shape, device=gs.device, dtype=torch.float32)
        self.num_actions = int(num_actions)
        self.actions, self.mu, self.sigma = z(T, n, num_actions), z(T, n, num_actions), z(T, n, num_actions)
        self.values, self.actions_log_prob, self.returns, self.advantages = z(T, n), z(T, n), z(T, n), z(T, n)
        self._moments = torch.zeros(2, device=gs.device, dtype=torch.float64)

    def add_policy(self, actions: torch.Tensor, values: torch.Tensor, log_prob: torch.Tensor, mu: torch.Tensor, sigma: torch.Tensor,
                   time_outs: Optional[torch.Tensor] = None, gamma: float = 0.99) -> None:
        """What the policy produced for the transition the last ``env.step()`` wrote (row ``step - 1``): rsl_rl's
        ``add_transitions`` for actions / values / actions_log_prob / mu / sigma, after ``rewards[t] += gamma * values *
        time_outs`` (PPO.process_env_step) when ``time_outs`` (this step's truncated flags) is given.  One launch."""
        if self.step < 1:
            raise RuntimeError("add_policy() follows the env.step() whose transition it completes")
        f = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()
        actions, mu, sigma = f(actions), f(mu), f(sigma)
        values, log_prob = f(values.reshape(-1)), f(log_prob.reshape(-1))
        self._ensure_policy_rows(actions.shape[-1])
        t, a = self.step - 1, self._pol_args
        a.num_envs, a.num_actions = self.env.num_envs, self.num_actions
        a.actions, a.values, a.log_prob, a.mu, a.sigma = (x.data_ptr() for x in (actions, values, log_prob, mu, sigma))
        a.actions_out, a.mu_out, a.sigma_out = (x.data_ptr() + t * x.stride(0) * 4 for x in (self.actions, self.mu, self.sigma))
        a.values_out, a.log_prob_out = (x.data_ptr() + t * x.stride(0) * 4 for x in (self.values, self.actions_log_prob))
        if time_outs is not None:
            if time_outs.dtype != torch.bool and time_outs.dtype != torch.uint8:
                time_outs = time_outs != 0
            time_outs = time_outs.contiguous()
            a.time_outs, a.reward_row, a.gamma = time_outs.data_ptr(), self.rewards.data_ptr() + t * self.rewards.stride(0) * 4, float(gamma)
        else:
            a.time_outs, a.reward_row, a.gamma = None, None, 0.0
        self._keep_pol = (actions, values, log_prob, mu, sigma, time_outs)
        self.env.backend.call("rollout_policy_write", a, owner=None)

    def compute_returns(self, last_values: torch.Tensor, gamma: float = 0.99, lam: float = 0.95, normalize: bool = True) -> None:
        """``returns`` / ``advantages`` of the finished rollout (rsl_rl ``RolloutStorage.compute_returns``; gamma / lam as
        examples/simple/train.py:41-47) — GAE over the T steps, then ``(adv - mean) / (std + 1e-8)`` over all T*N entries."""
        if self.values is None:
            raise RuntimeError("compute_returns() needs the value estimates: call add_policy() for every transition")
        last_values = last_values.reshape(-1).to(torch.float32).contiguous()
        g = self._gae_args
        g.num_envs, g.num_steps = self.env.num_envs, self.num_steps
        g.rewards, g.values, g.dones, g.last_values = self.rewards.data_ptr(), self.values.data_ptr(), self.dones.data_ptr(), last_values.data_ptr()
        g.gamma, g.lam = float(gamma), float(lam)
        g.returns, g.advantages, g.moments = self.returns.data_ptr(), self.advantages.data_ptr(), self._moments.data_ptr()
        g.normalize = 1 if normalize else 0
        self._keep_gae = last_values
        self.env.backend.call("gae", g, owner=None)

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
        if getattr(pol, "_window", False):
            from ._trace import Untraceable
            raise Untraceable("the rollout rows of a window-mode observation (a strided view) cannot be copied by the step's launch")
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

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, force: bool = False):
        """``force``: run the collective even in a group of ONE rank (the only RCCL group a one-GPU box can form: every call of the
        path executes, the sum over one rank is the identity — tests/test_learner.py)."""
        import torch.distributed as dist

        self.force = bool(force) and dist.is_available() and dist.is_initialized()

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
        if self.world == 1 and not self.force:
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
        if self.world == 1 and not self.force:
            return
        import torch.distributed as dist

        for p in self.params:
            dist.broadcast(p.data, src=src, group=self.group)
