"""
GenesisEnv — base vectorised environment (API mirror of genesis_forge/genesis_env.py:12-286).

Same public surface as the reference (``dt``, ``num_envs``, ``episode_length``,
``max_episode_length``, ``actions``, ``last_actions``, ``extras``, ``step``, ``reset`` …); the
per-env bookkeeping that the reference issues as separate torch ops is done by the native phase
kernels:

* ``step``  (genesis_env.py:181-205): ``episode_length += 1; last_actions <- actions; actions <- new``
  is part of ``gf_action_step`` (Phase A).
* ``reset`` (genesis_env.py:207-254): zeroing of actions / episode_length and the episode-length
  jitter are the "GenesisEnv.reset" section of ``gf_masked_reset`` (Phase R), driven by a mask
  instead of an index list so no host sync is needed when called from ``step``.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Literal, Optional

import torch

from . import _native as nat
from . import gs
from ._stats import LazyEpisodeLog, StepStats
from .spaces import Space

EnvMode = Literal["train", "eval", "play"]


class EntityViews:
    """Base-link state tensors of one entity for the current tick (world frame)."""

    __slots__ = ("pos", "quat", "lin_vel", "ang_vel")

    def __init__(self, pos, quat, lin_vel, ang_vel):
        self.pos, self.quat, self.lin_vel, self.ang_vel = pos, quat, lin_vel, ang_vel

    def fill(self, view: nat.GfEntityView) -> None:
        view.pos = self.pos.data_ptr()
        view.quat = self.quat.data_ptr()
        view.lin_vel = self.lin_vel.data_ptr()
        view.ang_vel = self.ang_vel.data_ptr()


class DoneIds:
    """The ascending index list of a step's done envs — what the reference gets from ``(terminated | truncated).nonzero()``
    (managed_env.py:308-310) — through ``gf_done_compact``: one small launch (two above 131 072 envs) writes the list and leaves the count in a pinned
    host word; the call itself waits for the stream (``wait``), so the one synchronisation of the step costs no second trip into
    the runtime.  torch's ``nonzero()`` costs an OR launch, a two-pass select, a device-to-host copy and an allocation for the
    same sync.

    Every compaction writes into a buffer of its own (the caching allocator hands back the block of a few steps ago): the list
    goes to user code — a ``reset(envs_idx)`` override may keep it — so it must never be rewritten by a later step, and a fresh
    block costs less than cloning the list out of a persistent one (an allocation against an allocation plus a copy launch)."""

    def __init__(self, num_envs: int):
        import ctypes as C

        dev = gs.device
        self.num_envs = num_envs
        self.count = torch.zeros(1, dtype=torch.int32)
        if dev.type == "cuda":
            self.count = self.count.pin_memory()   # the kernel stores into it through the host mapping
        self._count_word = C.c_int32.from_address(self.count.data_ptr())   # read without building a tensor
        self.scratch = torch.zeros((num_envs + 4095) // 4096 + 1, device=dev, dtype=torch.int32)
        self.args = nat.GfCompactArgs()
        self.args.num_envs = num_envs
        self.args.count_out, self.args.block_counts = self.count.data_ptr(), self.scratch.data_ptr()
        self.args.wait = 1 if dev.type == "cuda" else 0

    def __call__(self, backend, mask: torch.Tensor, mask2: Optional[torch.Tensor] = None) -> torch.Tensor:
        a = self.args
        ids = torch.empty(max(self.num_envs, 1), device=mask.device, dtype=torch.int64)
        a.ids_out = ids.data_ptr()
        a.mask, a.mask2 = mask.data_ptr(), (None if mask2 is None else mask2.data_ptr())
        tracer, backend.tracer = backend.tracer, None   # (never part of a recording: the list is taken where it is needed)
        try:
            backend.call("done_compact", a)   # returns when the stream has drained (the reference's nonzero() is a sync too)
        finally:
            backend.tracer = tracer
        return ids[:self._count_word.value]


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t if t.is_contiguous() else t.contiguous()


class GenesisEnv:
    """Base environment class (see module docstring; args as genesis_forge/genesis_env.py:52-59)."""

    action_space: Optional[Space] = None
    observation_space: Optional[Space] = None
    can_be_wrapped: bool = True

    def __init__(
        self,
        num_envs: int = 1,
        dt: float = 1 / 100,
        max_episode_length_sec: int | None = 10,
        max_episode_random_scaling: float = 0.0,
        extras_logging_key: str = "episode",
    ):
        self.dt = dt
        self.device = gs.device
        self.num_envs = num_envs
        self.scene = None
        self.robot = None
        self.terrain = None

        self.extras_logging_key = extras_logging_key
        self._extras: dict = {}
        self._extras[extras_logging_key] = LazyEpisodeLog()

        self._actions: Optional[torch.Tensor] = None
        self._last_actions: Optional[torch.Tensor] = None

        self.step_count: int = 0
        self.episode_length = torch.zeros((self.num_envs,), device=gs.device, dtype=torch.int32)
        self.max_episode_length: Optional[torch.Tensor] = None

        self._max_episode_length_sec = 0.0
        self._base_max_episode_length = None
        self._max_episode_random_scaling = max_episode_random_scaling
        if max_episode_length_sec and max_episode_length_sec > 0:
            self.max_episode_length = torch.zeros((self.num_envs,), device=gs.device, dtype=gs.tc_int)
            self.max_episode_length[:] = self.set_max_episode_length(max_episode_length_sec)

        # native-path state ----------------------------------------------------------------------
        self._stats: Optional[StepStats] = None
        self._in_step = False
        self._tick = 0                 # bumps whenever scene state may have changed (views cache key)
        self._views_cache: dict = {}
        self._rng_seed = 0x5EED
        self.env_offset = 0            # global index of local env 0 when envs are sharded over ranks (distributed.attach)
        self._rng_c = C.c_uint64(0)    # every stochastic native call takes a fresh stream id; a ctypes cell so that a recorded
        #                                step's native patch table (GfReplay.rng_stream) advances the very same counter
        self._draws: dict = {}         # parity mode: {"name": tensor of U[0,1)} consumed by the next call
        self._done_mask: Optional[torch.Tensor] = None
        self._trace = None
        self._trace_epoch = 0
        self._recorder = None
        self._last_signature = None
        #: snapshot + write-back for a scene with Genesis' public surface only (fresh getter tensors, envs_idx setters); None for
        #: a scene whose state tensors are persistent and shared (``gf_static_buffers``); set by build()
        self._adapter = None
        self._done_ids_native: Optional[DoneIds] = None
        self._done_ids_cache = None

    """
    Properties (genesis_env.py:95-148)
    """

    @property
    def unwrapped(self):
        return self

    @property
    def max_episode_length_sec(self) -> int | None:
        return self._max_episode_length_sec

    @property
    def extras(self) -> dict:
        return self._extras

    @property
    def actions(self) -> torch.Tensor:
        return self._actions

    @property
    def last_actions(self) -> torch.Tensor:
        return self._last_actions

    @property
    def num_actions(self) -> int:
        if self.action_space is not None:
            return self.action_space.shape[0]
        return 0

    @property
    def num_observations(self) -> int:
        if self.observation_space is not None:
            return self.observation_space.shape[0]
        return 0

    @property
    def max_episode_length_steps(self) -> int | None:
        return self._base_max_episode_length

    """
    Utilities
    """

    def set_max_episode_length(self, max_episode_length_sec: int) -> int:
        """genesis_env.py:153-165"""
        self._max_episode_length_sec = max_episode_length_sec
        self._base_max_episode_length = math.ceil(max_episode_length_sec / self.dt)
        self.invalidate_trace()   # the reset descriptor of a recorded step carries the base length
        return self._base_max_episode_length

    def seed(self, seed: int) -> None:
        """Seed of the in-kernel Philox generator (command resampling, reset jitter, observation noise)."""
        self._rng_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.invalidate_trace()

    def invalidate_trace(self) -> None:
        """Drop the recorded step (something its frozen descriptors depend on has changed)."""
        if not hasattr(self, "_trace_epoch"):
            return  # still inside __init__: nothing has been recorded
        st = self._stats
        if self._trace is not None and st is not None and getattr(st, "ring", None) is not None and (st.group is None or st.reduce_every > 1):
            # (batched logging reduction, reduce_every > 1: this closes the open batch — its newest row is folded explicitly —
            # before the recorded step goes away: a collective, like everything that invalidates a recorded step on one rank must
            # happen on all of them)
            last = st.end_recording()
            rm = getattr(self, "managers", {}).get("reward") if hasattr(self, "managers") else None
            if rm is not None:
                rm._apply_reset_stats(last)
        self._trace = None
        self._trace_epoch += 1
        self._last_signature = None
        if getattr(self, "_soft_dirty", None):
            self._soft_dirty.clear()

    @property
    def _rng_stream(self) -> int:
        return self._rng_c.value

    @_rng_stream.setter
    def _rng_stream(self, v: int) -> None:
        self._rng_c.value = v

    def next_stream(self) -> int:
        v = self._rng_c.value + 1
        self._rng_c.value = v
        return v

    def set_draws(self, **draws: torch.Tensor) -> None:
        """Parity mode: supply the U[0,1) draws the next stochastic phase calls consume instead of Philox.
        Keys: ``episode_length`` [N], ``dof_reset`` [N,D], ``command:<i>`` [N,R], ``command_reset:<i>`` [N,R],
        ``obs:<name>`` [N,O]."""
        for k, v in draws.items():
            self._draws[k] = None if v is None else _f32c(v.to(gs.device))
        self.invalidate_trace()

    def take_draws(self, key: str) -> Optional[torch.Tensor]:
        return self._draws.pop(key, None)

    # -- native plumbing ---------------------------------------------------------------------------
    @property
    def backend(self) -> nat.Backend:
        return nat.get_backend()

    @property
    def stats(self) -> StepStats:
        if self._stats is None:
            self._stats = StepStats(gs.device)
        return self._stats

    def entity_views(self, entity) -> EntityViews:
        """World-frame base state of ``entity`` for this tick, fetched once (the reference re-fetches it in
        every term: utils.py:23-24,37-38,51-55)."""
        key = id(entity)
        hit = self._views_cache.get(key)
        ad = self._adapter
        if ad is not None and not hasattr(entity, "gf_views"):
            # Genesis-shaped scene: each getter is called once per tick (the snapshot); the masked reset of the tick writes the
            # post-reset rows into these very tensors, so they stay valid until the scene steps again
            if hit is not None and hit[0] == ("snap", ad.epoch):
                return hit[1]
            v = EntityViews(ad.base(entity, "pos"), ad.base(entity, "quat"), ad.base(entity, "vel"), ad.base(entity, "ang"))
            self._views_cache[key] = (("snap", ad.epoch), v)
            return v
        if hit is not None and hit[0] == self._tick:
            return hit[1]
        if hasattr(entity, "gf_views"):
            v = entity.gf_views()  # synthetic scene: persistent buffers, zero copy
        else:
            v = EntityViews(_f32c(entity.get_pos()), _f32c(entity.get_quat()), _f32c(entity.get_vel()), _f32c(entity.get_ang()))
        self._views_cache[key] = (self._tick, v)
        return v

    def done_ids(self, mask: torch.Tensor, mask2: Optional[torch.Tensor] = None, own: bool = False) -> torch.Tensor:
        """Ascending indices of the envs whose ``mask`` (or ``mask2``) is set — see :class:`DoneIds`.  The list is the caller's to
        keep (``own`` is accepted for older call sites: every list is in a buffer of its own)."""
        if self._done_ids_native is None:
            self._done_ids_native = DoneIds(self.num_envs)
        # only ever asked for the termination masks of the current step, which are written once per step: a second request within
        # the step — the setters' push, then a user manager's reset(ids) — takes the list the first one compacted
        key = (self.step_count, self._in_step, mask.data_ptr(), None if mask2 is None else mask2.data_ptr())
        hit = self._done_ids_cache
        if hit is not None and hit[0] == key:
            return hit[1]
        ids = self._done_ids_native(self.backend, mask, mask2)
        self._done_ids_cache = (key, ids, mask, mask2, None)
        return ids

    def invalidate_views(self) -> None:
        self._tick += 1

    def scene_stepped(self) -> None:
        """``scene.step()`` has run (or the scene was changed from outside): state read before it is stale."""
        self._tick += 1
        if self._adapter is not None:
            self._adapter.invalidate()

    """
    Operations
    """

    def build(self) -> None:
        """genesis_env.py:171-179"""
        assert self.scene is not None, "The scene must be constructed and assigned to the <env>.scene attribute before building."
        self.scene.build(n_envs=self.num_envs)
        if not getattr(self.scene, "gf_static_buffers", False):
            from ._scene_adapter import SceneAdapter
            self._adapter = SceneAdapter(self)

    def _begin_step(self) -> None:
        self._extras = {}
        self._extras[self.extras_logging_key] = LazyEpisodeLog()
        self.step_count += 1
        self._in_step = True
        self.stats.clear(self.backend)

    def _end_step(self) -> None:
        log = self._extras.get(self.extras_logging_key)
        if isinstance(log, LazyEpisodeLog):
            log.attach(self.stats.snapshot())
        self._in_step = False

    def _ensure_action_buffers(self, actions: torch.Tensor) -> None:
        if self._actions is None:
            self._actions = torch.zeros_like(actions, device=gs.device, dtype=gs.tc_float).contiguous()
            self._last_actions = torch.zeros_like(self._actions)

    def _bookkeep(self, actions: torch.Tensor) -> None:
        """episode_length += 1; last_actions <- actions; actions <- new (genesis_env.py:196-203) as one launch,
        used when no action manager fuses it into Phase A."""
        actions = _f32c(actions)
        self._ensure_action_buffers(actions)
        D = actions.shape[1]
        if getattr(self, "_bk_consts", None) is None or self._bk_consts[0].numel() != D:
            one = torch.ones(D, device=gs.device)
            zero = torch.zeros(D, device=gs.device)
            self._bk_consts = (one, zero, torch.full((D,), -torch.inf, device=gs.device), torch.full((D,), torch.inf, device=gs.device),
                               torch.empty_like(self._actions))
        one, zero, lo, hi, scratch = self._bk_consts
        a = self.__dict__.get("_bk_args")   # (one descriptor for the env's life: a recorded step patches it in place)
        if a is None:
            a = self._bk_args = nat.GfActionArgs()
        a.num_envs, a.num_dofs, a.mode, a.check_finite = self.num_envs, D, nat.GF_ACTION_POSITION, 0
        a.actions_in = actions.data_ptr()
        a.scale, a.offset, a.clip_lo, a.clip_hi = one.data_ptr(), zero.data_ptr(), lo.data_ptr(), hi.data_ptr()
        a.env_actions, a.env_last_actions = self._actions.data_ptr(), self._last_actions.data_ptr()
        a.episode_length = self.episode_length.data_ptr()
        a.targets = scratch.data_ptr()
        a.stats = None
        self.backend.call("action_step", a)

    def step(self, actions: torch.Tensor):
        """genesis_env.py:181-205"""
        self._begin_step()
        self._bookkeep(actions)
        return None, None, None, None, self._extras

    # -- reset -------------------------------------------------------------------------------------
    def _ids_to_mask(self, envs_idx) -> torch.Tensor:
        hit = self._done_ids_cache
        if hit is not None and envs_idx is hit[1] and hit[0][0] == self.step_count:   # (of THIS step: the masks are rewritten every step)
            # the list this step's done_ids() compacted out of the termination masks, handed back (a user manager's reset(ids), a
            # reset() override calling super().reset(ids)): its mask IS those masks — one OR per step at most, instead of a zeros +
            # an index_put per manager that turns the list back into a mask for its masked launch
            if hit[4] is None:
                m1, m2 = hit[2], hit[3]
                self._done_ids_cache = hit = hit[:4] + ((m1 if m2 is None else (m1 | m2)),)
            return hit[4]
        mask = torch.zeros(self.num_envs, dtype=torch.bool, device=gs.device)
        if envs_idx is None:
            mask[:] = True
        else:
            idx = torch.as_tensor(envs_idx, device=gs.device, dtype=torch.long)
            if idx.numel() > 0:
                mask[idx] = True
        return mask

    def _fill_env_reset(self, a: nat.GfResetArgs) -> None:
        """The GenesisEnv.reset section of the fused reset (genesis_env.py:233-252)."""
        if self.step_count == 0 and self.action_space is not None and self._actions is None:
            self._actions = torch.zeros((self.num_envs, self.action_space.shape[0]), device=gs.device, dtype=gs.tc_float)
            self._last_actions = torch.zeros_like(self._actions)
        a.num_envs = self.num_envs
        if self._actions is not None:
            a.num_dofs = self._actions.shape[1]
            a.env_actions = self._actions.data_ptr()
            a.env_last_actions = self._last_actions.data_ptr()
        a.episode_length = self.episode_length.data_ptr()
        if self._max_episode_random_scaling > 0.0 and self._base_max_episode_length is not None and self.max_episode_length is not None:
            a.max_episode_length = self.max_episode_length.data_ptr()
            a.base_max_episode_length = int(self._base_max_episode_length)
            a.max_random_scaling = float(self._base_max_episode_length * self._max_episode_random_scaling)
            d = self.take_draws("episode_length")
            self._keep = (d,)
            a.len_draws = None if d is None else d.data_ptr()
        a.seed = self._rng_seed
        a.stream = self.next_stream()
        a.env_offset = self.env_offset

    def _reset_masked(self, mask: torch.Tensor, mask2: Optional[torch.Tensor] = None) -> None:
        a = nat.GfResetArgs()
        a.mask = mask.data_ptr()
        a.mask2 = None if mask2 is None else mask2.data_ptr()
        self._fill_env_reset(a)
        a.stats = self.stats.ptr
        self.backend.call("masked_reset", a)

    def reset(self, envs_idx: list[int] = None):
        """genesis_env.py:207-254"""
        self._reset_masked(self._ids_to_mask(envs_idx))
        self.invalidate_views()
        return None, self.extras

    def get_observations(self) -> torch.Tensor:
        """genesis_env.py:256-282"""
        if self.observation_space is not None:
            return torch.zeros((self.num_envs, self.observation_space.shape[0]), device=gs.device, dtype=gs.tc_float)
        return None

    def close(self):
        pass
