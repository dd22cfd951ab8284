"""
RewardManager — API mirror of genesis_forge/managers/reward_manager.py.

``step`` (:166-195) is one ``gf_reward_step`` launch: the weighted left-fold over the cfg order,
``episode_seconds += dt`` and the per-term episode sums all happen in the kernel.  ``reset`` (:197-222)
is the RewardManager section of the fused masked reset; the episode means it logs are accumulated
on device and surface in ``extras["episode"]`` lazily (no ``.item()`` per term).
"""
from __future__ import annotations

from functools import partial
from typing import Any, Callable, Optional, TypedDict

import torch

from .. import _native as nat
from .. import gs
from .._stats import GroupRingSnapshot, RingSnapshot
from ._program import RewardProgram, refresh_terms, same_structure, spec_of
from .base import BaseManager, LiveAttr
from .config import RewardConfigItem


class RewardConfig(TypedDict):
    fn: Callable[..., torch.Tensor]
    params: dict[str, Any]
    weight: float


class RewardManager(BaseManager):
    logging_enabled = LiveAttr("logging_enabled")   # checked on every step / reset in the reference (reward_manager.py:191,200)

    """Calculates and logs the rewards (ctor as reward_manager.py:89-118)."""

    _fused_reset = True

    def __init__(self, env, cfg: dict[str, RewardConfig], logging_enabled: bool = True, logging_tag: str = "Rewards"):
        super().__init__(env, type="reward")
        self.logging_enabled = logging_enabled
        self.logging_tag = logging_tag
        if len(cfg) > nat.GF_MAX_TERMS:
            raise ValueError(f"RewardManager supports at most {nat.GF_MAX_TERMS} terms")

        self.cfg: dict[str, RewardConfigItem] = {}
        for name, c in cfg.items():
            self.cfg[name] = RewardConfigItem(c, env, on_dirty=self._mark_dirty)

        N, T = env.num_envs, max(len(self.cfg), 1)
        self._reward_buf = torch.zeros((N,), device=gs.device, dtype=gs.tc_float)
        self._episode_seconds = torch.zeros((N,), device=gs.device, dtype=gs.tc_float)
        # SoA [T, N]: each term's running episode sum is one coalesced column; the reference's
        # per-name dict (reward_manager.py:114-118) is a dict of row views of it.
        self._episode_sums = torch.zeros((T, N), device=gs.device, dtype=gs.tc_float)
        self._episode_mean: dict[str, float] = dict()
        self._episode_data: dict[str, torch.Tensor] = {name: self._episode_sums[i] for i, name in enumerate(self.cfg.keys())}
        self._program: Optional[RewardProgram] = None
        self._log_names = None
        self._pending_names = None
        self._dirty = True
        self._pending: list = []  # snapshots with unreduced episode means (for last_episode_mean_reward)
        self._weights_only = False  # every edit since the last compile was a `.weight` assignment (_refresh_in_place)

    def _mark_dirty(self, soft: bool = False, what: Optional[str] = None):
        """``soft`` (a weight or a param VALUE was assigned — a curriculum): nothing is dropped; the next step compiles the table again and,
        when only numbers differ, writes them into the descriptor it already has — the one a recorded step froze
        (_compile, ManagedEnvironment._refresh_soft).  Anything else drops the recorded step right away."""
        if not self._dirty:
            self._weights_only = True
        self._weights_only = self._weights_only and soft and what == "_weight"   # (since the table was last compiled)
        self._dirty = True
        self._log_names = None
        env = self.env
        if soft and hasattr(env, "_soft_dirty"):
            env._soft_dirty.add(self)
        elif hasattr(env, "invalidate_trace"):
            env.invalidate_trace()

    def _refresh_in_place(self) -> bool:
        """A weight / param edit while a recorded step exists: True when the compile kept the descriptor (only numbers differed);
        otherwise the structure changed (a weight to or from zero, a stateful term's buffer …) and the compile dropped the recording."""
        old = self._program
        if old is None or not self.enabled:
            return False
        if self._weights_only:
            # only `.weight` assignments since the last compile (a weight annealed every step): the rows' weights are written straight
            # into the table — unless a weight went to or from zero, which changes the table's rows (reward_manager.py:181-182)
            a, dt = old.args, self.env.dt
            live = [(row, cfg.weight) for row, cfg in enumerate(self.cfg.values()) if cfg.weight != 0]
            if [row for row, _w in live] == [a.terms[k].row for k in range(old.n)]:
                for k, (_row, w) in enumerate(live):
                    a.terms[k].w = w * dt
                self._dirty = False
                return True
        self._compile()
        return self._program is old

    @property
    def rewards(self) -> torch.Tensor:
        return self._reward_buf

    @property
    def episode_data(self) -> dict[str, torch.Tensor]:
        return self._episode_data

    # -- helpers ------------------------------------------------------------------------------------
    def last_episode_mean_reward(self, name: str, before_weight: bool = True) -> float:
        """reward_manager.py:138-153.  Reading it drains the pending statistics snapshots (the only place this
        manager ever waits for the device)."""
        self._drain_pending()
        st = self.env.stats.read_last_reset() if self.env._trace is not None else None
        if st is not None:  # recorded steps keep this on the device; read it on demand
            self._apply_reset_stats(st)
        rew = self._episode_mean.get(name, 0.0)
        if before_weight:
            rew /= self.cfg[name].weight
        return rew

    def _apply_reset_stats(self, st) -> None:
        if st is not None and st.reset_count > 0:
            for row, (name, cfg) in enumerate(self.cfg.items()):
                if cfg.weight != 0:
                    self._episode_mean[name] = float(st.reward_episode_sum[row] / st.reset_count)

    def _drain_pending(self, keep_last: int = 0):
        while len(self._pending) > keep_last:
            snap, names, n_total = self._pending.pop(0)
            st = snap.wait()
            if st.reset_count > 0:
                for row, name in names:
                    self._episode_mean[name] = float(st.reward_episode_sum[row] / st.reset_count)

    # -- operations -----------------------------------------------------------------------------------
    def build(self):
        for cfg in self.cfg.values():
            cfg.build()

    def _compile(self):
        env = self.env
        old = self._program
        prog = RewardProgram(env)
        dt = env.dt
        for row, (name, cfg) in enumerate(self.cfg.items()):
            if cfg.weight == 0:  # reward_manager.py:181-182: skipped, not multiplied by zero
                continue
            fn, params = cfg.fn, cfg.params
            spec = spec_of(fn, env, params)
            prog.add(spec, (lambda fn=fn, params=params: fn(env, **params)), cfg.weight * dt, row)
            if spec is not None and spec.after is not None:
                prog.after.append(self._mark_dirty)  # stateful term flipped its first-call flag
        a = prog.args
        a.mode = nat.GF_REWARD_MODE_STEP
        a.logging_enabled = 1 if self.logging_enabled else 0
        a.reward = self._reward_buf.data_ptr()
        a.episode_sums = self._episode_sums.data_ptr()
        a.episode_seconds = self._episode_seconds.data_ptr()
        prog.manager = self
        if old is not None and same_structure(old, prog) and old.args.logging_enabled == a.logging_enabled:
            # only numbers differ (a curriculum edit): they go into the descriptor that exists — a recorded step that froze it goes on,
            # and a recording in progress keeps seeing the same descriptor from step to step
            refresh_terms(old, prog)
            prog = old
        elif old is not None and getattr(env, "_trace", None) is not None:
            env.invalidate_trace()   # another structure under a recorded step
        self._program = prog
        self._dirty = False

    def step(self) -> torch.Tensor:
        """reward_manager.py:166-195"""
        if not self.enabled:
            return self._reward_buf
        if self._dirty:
            self._compile()
        self._program.launch()
        return self._reward_buf

    def _log_mask(self) -> int:
        m = 0
        for row, cfg in enumerate(self.cfg.values()):
            if cfg.weight != 0:
                m |= 1 << row
        return m

    def _fill_reset(self, a: nat.GfResetArgs) -> None:
        a.episode_seconds = self._episode_seconds.data_ptr()
        if self.enabled and self.logging_enabled:
            a.episode_sums = self._episode_sums.data_ptr()
            a.num_reward_terms = len(self.cfg)
            a.reward_log_mask = self._log_mask()
            a.reward_logging = 1
            self._register_log()

    def _register_log(self):
        """Queue the "Rewards / <name>" entries for this reset event (reward_manager.py:202-216)."""
        env = self.env
        log = env._extras[env.extras_logging_key]
        names = self._log_names
        if names is None:  # rebuilt only after a weight / param mutation (_mark_dirty)
            names = self._log_names = [(row, name) for row, (name, cfg) in enumerate(self.cfg.items()) if cfg.weight != 0]
        if hasattr(log, "add_filler"):
            log.add_filler(partial(self._fill_log, names))
        self._pending_names = names

    def _fill_log(self, names, st, out):
        if st.reset_count > 0:
            tag = self.logging_tag
            for row, name in names:
                mean = st.reward_episode_sum[row] / st.reset_count
                self._episode_mean[name] = float(mean)
                out[f"{tag} / {name}"] = torch.tensor(mean, dtype=torch.float32)

    def _note_snapshot(self, snap):
        """Called by the env after the step's snapshot exists, so curricula can read means later."""
        names = self._pending_names
        if type(snap) is RingSnapshot or type(snap) is GroupRingSnapshot:
            self._pending_names = None  # recorded step (single process / batched group ring): the device keeps the last reset statistics
            return
        if names:
            self._pending.append((snap, names, self.env.num_envs))
            self._pending_names = None
            if len(self._pending) > 56:
                # one batched read-back per ~48 steps (the statistics ring holds 64): keeps _episode_mean exact without a
                # per-step sync
                self._drain_pending(keep_last=8)

    def reset(self, envs_idx: list[int] | None = None):
        """reward_manager.py:197-222 (standalone form)."""
        env = self.env
        a = nat.GfResetArgs()
        a.num_envs = env.num_envs
        mask = env._ids_to_mask(envs_idx)
        a.mask = mask.data_ptr()
        self._fill_reset(a)
        a.stats = env.stats.ptr
        env.backend.call("masked_reset", a)
