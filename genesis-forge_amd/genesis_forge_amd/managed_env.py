"""
ManagedEnvironment — orchestrator of the manager step pipeline (API mirror of
genesis_forge/managed_env.py:20-398).

Phase order is the reference's (managed_env.py:274-334):
    A  action.step(actions)      → gf_action_step  (+ GenesisEnv.step bookkeeping fused in)
    P  scene.step()              → Genesis / synthetic scene (out of scope)
    B1 entity.step()   B2 contact.step() → gf_contact_step per manager
    B3 termination.step()        → gf_termination_step
    B4 reward.step()             → gf_reward_step
    B5 command.step()            → gf_command_step per manager
    R  reset of done envs        → gf_masked_reset (+ gf_command_step masked)   — no nonzero() sync
    O  get_observations()        → gf_observe per ObservationManager
A step is ~8 kernel launches and zero host syncs (reference: 172 aten ops, ≥6 syncs).
Quirks q5/q6 hold: commands are resampled from the pre-reset episode_length, rewards come from the
terminal state, observations from the post-reset state.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Optional

import torch
from functools import partial

from . import _native as nat
from . import _trace
from . import gs
from ._stats import LazyEpisodeLog
from .genesis_env import GenesisEnv
from .managers.action import PositionActionManager
from .managers.base import BaseManager, ManagerType

try:  # pragma: no cover - tensordict is absent in this image
    from tensordict import TensorDict as _TensorDict

    def _obs_dict():
        return _TensorDict({}, device=gs.device)
except Exception:

    class ObservationDict(dict):
        """Plain-dict stand-in for ``TensorDict({}, device=…)`` (managed_env.py:287)."""

        def to(self, *a, **k):
            return ObservationDict({key: v.to(*a, **k) for key, v in self.items()})

    def _obs_dict():
        return ObservationDict()


def _most_derived_is_ours(m, method: str) -> bool:
    """True when the most-derived implementation of ``method`` on ``m`` comes from this package (not a user subclass)."""
    for klass in type(m).__mro__:
        if method in klass.__dict__:
            return klass.__module__.startswith(__package__ + ".")
    return True


def _most_derived_reset_is_ours(m: BaseManager) -> bool:
    return _most_derived_is_ours(m, "reset")


class ManagedEnvironment(GenesisEnv):
    """An environment whose step/reset logic is supplied by registered managers (ctor as managed_env.py:113-127)."""

    def __init__(self, num_envs: int = 1, dt: float = 1 / 100, max_episode_length_sec: int | None = 10,
                 max_episode_random_scaling: float = 0.0, extras_logging_key: str = "episode"):
        super().__init__(num_envs=num_envs, dt=dt, max_episode_length_sec=max_episode_length_sec,
                         max_episode_random_scaling=max_episode_random_scaling, extras_logging_key=extras_logging_key)
        self.managers: dict = {"contact": [], "entity": [], "command": [], "terrain": [], "action": None, "observation": [],
                               "reward": None, "termination": None}
        self._action_space = None
        self._observation_space = None
        self._reward_buf = torch.zeros((self.num_envs,), device=gs.device, dtype=gs.tc_float)
        self._terminated_buf = torch.zeros((self.num_envs,), device=gs.device, dtype=gs.tc_bool)
        self._truncated_buf = torch.zeros((self.num_envs,), device=gs.device, dtype=gs.tc_bool)
        self._reset_args = nat.GfResetArgs()
        self._last_images = None  # descriptor images of the previous recorded ordinary step (Genesis-shaped scene)
        self._untraceable: Optional[str] = None   # why the last attempt to record the step was refused
        self._no_trace_epoch = -1
        self._partition_cache = None
        self._program_pending = None   # a static program of this config being compiled in a child process (_programs.Pending)
        self._soft_dirty: set = set()   # managers whose term-table NUMBERS changed since the last step (a recorded step refreshes them in place)
        self._program_info: Optional[dict] = None   # what became of it: signature, plugin path, compile seconds (or the error)
        self._program_info_tail: Optional[dict] = None   # the same for the observation-only launch of a reset()-override tail
        #: "off" / "sync" / "async": compile a static program of the fused post-physics kernel for this config's structure when no
        #: built-in one matches (None: by size, see _programs.mode_for; GF_JIT overrides)
        self.jit_programs: Optional[str] = None
        self._done_ids = None     # the index list of this step's done envs while a user reset() override holds it (reset() recognises it)
        self._tail_trace = None   # the recorded step whose Python tail is running (its reset / observation segments replay natively)
        #: record the step and replay it through gf_run_ops when possible (see _trace.py); GF_NO_TRACE=1 disables
        self.trace_enabled = os.environ.get("GF_NO_TRACE", "0") != "1"
        #: replace the recorded post-physics phases by the fused gf_post_physics_step launch; GF_NO_FUSE=1 disables
        self.fuse_post_physics = os.environ.get("GF_NO_FUSE", "0") != "1"

    # -- spaces (managed_env.py:156-194) ------------------------------------------------------------
    @property
    def action_space(self):
        if self.managers["action"] is not None:
            return self.managers["action"].action_space
        return self._action_space

    @action_space.setter
    def action_space(self, action_space):
        self._action_space = action_space

    @property
    def observation_space(self):
        if len(self.managers["observation"]) > 0:
            for obs in self.managers["observation"]:
                if obs.name == "policy":
                    return obs.observation_space
            return self.managers["observation"][0].observation_space
        return self._observation_space

    @observation_space.setter
    def observation_space(self, observation_space):
        self._observation_space = observation_space

    # -- managers -------------------------------------------------------------------------------------
    def add_manager(self, manager_type: ManagerType, manager: BaseManager):
        """managed_env.py:200-220"""
        if manager_type not in self.managers:
            raise ValueError(f"'{manager_type}' is not a valid manager type.")
        if isinstance(self.managers[manager_type], list):
            self.managers[manager_type].append(manager)
        elif self.managers[manager_type] is None:
            self.managers[manager_type] = manager
        else:
            raise ValueError(f"Manager type '{manager_type}' already has a manager, and an environment cannot have multiple {manager_type} managers.")
        self.invalidate_trace()

    def _all_managers(self) -> list:
        out = []
        if self.managers["action"] is not None:
            out.append(self.managers["action"])
        out += self.managers["entity"] + self.managers["contact"]
        if self.managers["termination"] is not None:
            out.append(self.managers["termination"])
        if self.managers["reward"] is not None:
            out.append(self.managers["reward"])
        out += self.managers["command"] + self.managers["observation"]
        return out

    # -- operations -----------------------------------------------------------------------------------
    def config(self):
        """Override this method and create all your managers here (managed_env.py:226-247)."""

    def build(self):
        """managed_env.py:249-272 (same build order)."""
        super().build()
        self.config()
        for m in self.managers["terrain"]:
            m.build()
        if self.managers["action"] is not None:
            self.managers["action"].build()
        for m in self.managers["contact"]:
            m.build()
        if self.managers["termination"] is not None:
            self.managers["termination"].build()
        if self.managers["reward"] is not None:
            self.managers["reward"].build()
        for m in self.managers["command"]:
            m.build()
        for m in self.managers["entity"]:
            m.build()
        for m in self.managers["observation"]:
            m.build()

    def step(self, actions: torch.Tensor):
        """managed_env.py:274-334"""
        if actions.dtype != torch.float32 or not actions.is_contiguous():
            actions = actions.to(torch.float32).contiguous()
        tr = self._trace
        if self._soft_dirty:
            tr = self._refresh_soft()
        if tr is not None:
            if tr.epoch == self._trace_epoch and not self._draws:
                if tr.fresh():
                    if self._program_pending is not None and self._program_pending.poll():
                        self._program_pending = None   # the config's own static program is registered: the next launch matches it
                    return tr.replay(actions)
                tr = None
                self.invalidate_trace()  # a manager method called between steps went through one of its descriptors
            else:
                self._trace = None
        if not self.trace_enabled or self._draws:
            return self._step_ordinary(actions)
        if self._no_trace_epoch == self._trace_epoch:
            return self._step_ordinary(actions)   # recording was tried in this configuration and refused (self._untraceable)
        rec = _trace.Recorder(images=self._adapter is not None)
        backend = self.backend
        backend.tracer = rec
        epoch = self._trace_epoch
        try:
            out = self._step_ordinary(actions)
        finally:
            backend.tracer = None
        if epoch == self._trace_epoch:  # nothing was invalidated while the step ran
            sig = rec.signature()
            if sig == self._last_signature and _trace.traceable(self, rec.tail_python):
                try:
                    self._trace = _trace.StepTrace(self, rec.calls, rec.tail_python, rec.tail, images=(self._last_images, rec.images))
                    if self._trace.post_refs is not None:
                        from . import _programs
                        _programs.on_recorded(self)   # a config no built-in program matches gets its own (GF_JIT, _programs.py)
                except _trace.Untraceable as why:
                    self._untraceable, self._no_trace_epoch = str(why), self._trace_epoch
            self._last_signature = sig
            self._last_images = rec.images
        return out

    def _refresh_soft(self):
        """Weights / param values were assigned since the last step (a curriculum).  With a recorded step: every manager concerned
        compiles its table again and writes the new numbers into the descriptor the recording froze; one whose STRUCTURE changed drops
        the recording (its _compile does).  Without one: nothing to do here — the managers are dirty and compile at their phase."""
        todo, self._soft_dirty = list(self._soft_dirty), set()
        if self._trace is None:
            return None
        ok = False
        try:
            ok = all([m._refresh_in_place() for m in todo])
        finally:
            if not ok and self._trace is not None:   # (also when a compile raised — a bad param value: never go on with the old table)
                self.invalidate_trace()
        tr = self._trace
        if tr is not None and any(m in self.managers["observation"] for m in todo):
            # an item's scale / noise switched on or off is part of a run-time compiled program's signature (not of the table layout):
            # the launch falls back to the interpreter by itself; give the new structure its own program, as a re-recording would
            from . import _programs
            if tr.post_refs is not None:
                _programs.on_recorded(self)
            if getattr(tr, "_tail_refs", None) is not None and tr.tail_seg:
                _programs.on_recorded(self, tr._tail_refs)
        return tr

    def _begin_step_light(self) -> None:
        """_begin_step without the statistics clear (a recorded step carries it as its first op)."""
        self._extras = {self.extras_logging_key: LazyEpisodeLog(), "observations": _obs_dict()}
        self.step_count += 1

    def _finish_step_light(self, snap) -> None:
        self._extras[self.extras_logging_key].attach(snap)
        rm = self.managers["reward"]
        if rm is not None:
            rm._note_snapshot(snap)

    def _after_masked_reset_traced(self) -> None:
        tm = self.managers["termination"]
        for em in self.managers["entity"]:
            em._after_fused_reset(tm._terminated_buf, tm._truncated_buf)
        self.invalidate_views()
        ad = self._adapter
        if ad is not None:   # Genesis-shaped scene: the simulator learns the reset rows through its envs_idx setters
            if self._done_ids is not None:
                ad.push(self._done_ids)   # (a reset() override already paid for the index list)
            else:
                ad.push_done(tm._terminated_buf, tm._truncated_buf)

    def _step_ordinary(self, actions: torch.Tensor):
        self._begin_step()
        self.extras["observations"] = _obs_dict()

        # A: actions (+ env bookkeeping) and the simulation step
        am = self.managers["action"]
        if am is not None and isinstance(am, PositionActionManager) and type(am).step is PositionActionManager.step and not am._user_handler():
            am.step(actions, _fuse_env=True)
        else:
            # a user-defined action manager class (its own step() / handle_actions()): the env's bookkeeping as a launch of its own,
            # then the user's code — while the step is being recorded, as user code between native phases (_manager_step)
            self._bookkeep(actions)
            if am is not None:
                rec = self.backend.tracer
                if rec is not None and isinstance(am, PositionActionManager):
                    from .managers._program import call_untraced
                    self._step_actions = actions
                    rec.python(self._user_action_step)
                    call_untraced(self, self._user_action_step)
                else:
                    am.step(actions)
        self.scene.step()
        self.scene_stepped()

        for m in self.managers["entity"]:
            self._manager_step(m)
        for m in self.managers["contact"]:
            self._manager_step(m)

        truncated, terminated = self._truncated_buf, self._terminated_buf
        tm = self.managers["termination"]
        if tm is not None:
            terminated, truncated = self._manager_step(tm)

        rewards = self._reward_buf
        if self.managers["reward"] is not None:
            rewards = self._manager_step(self.managers["reward"])

        for m in self.managers["command"]:
            self._manager_step(m)

        if tm is not None:
            if type(self).reset is not ManagedEnvironment.reset and self.backend.tracer is not None:
                self.backend.tracer.cut_tail()  # a user reset(): the rest of the step stays Python in the recorded step
            self._reset_done(terminated, truncated)

        rec = self.backend.tracer
        if rec is not None and not rec.tail_python and type(self).get_observations is not ManagedEnvironment.get_observations:
            from .managers._program import call_untraced
            rec.python(self._user_get_observations)
            call_untraced(self, self._user_get_observations)
            obs = self._step_obs
        else:
            obs = self.get_observations()
        ro = getattr(self, "_rollout", None)
        if ro is not None and tm is not None:
            # the RL library's rollout rows (learner.RolloutStorage): written from the manager-owned buffers of this step
            pol = next((m for m in self.managers["observation"] if m.name == ro.obs_name), None)
            ro.write(pol._last_out if pol is not None else obs, rewards, terminated, truncated)
        self._end_step()
        return obs, rewards, terminated, truncated, self.extras

    def _user_action_step(self) -> None:
        """``step(actions)`` of a user-defined action manager class with this step's actions (a recorded step calls it at its place)."""
        self.managers["action"].step(self._step_actions)

    def _manager_step(self, m) -> None:
        """``m.step()`` — for a user-defined manager class while the step is being recorded: as user code between native phases
        (the recording keeps its place and calls it again there; launches it makes itself belong to it, not to the recording)."""
        rec = self.backend.tracer
        if rec is None or _most_derived_is_ours(m, "step"):
            return m.step()
        from .managers._program import call_untraced
        rec.python(m.step)
        return call_untraced(self, m.step)

    def _indexed_reset(self, indexed: list, mask: torch.Tensor, mask2: Optional[torch.Tensor]) -> None:
        """``reset(ids)`` of the managers that need an index list (user-defined classes, Python on_reset entries), for the done
        envs of this step: the one ``nonzero()`` the reference pays too (managed_env.py:308-310)."""
        ids = self.done_ids(mask, mask2, own=True)   # (user code may keep the list: a tensor of its own)
        if ids.numel() > 0:
            for m in indexed:
                m.reset(ids)
            self._after_indexed_reset(indexed)

    def _after_indexed_reset(self, indexed: list) -> None:
        """Genesis-shaped scene: the reset(ids) of an action / entity manager goes through the simulator's envs_idx setters (joint
        positions, base pose — also a Python on_reset entry); this tick's snapshot does not know, so what reads state from here on
        fetches it again, as the reference's getters do.  (Such a step is never a recorded one: _trace.traceable.)"""
        ad = self._adapter
        if ad is not None and any(m is self.managers["action"] or m in self.managers["entity"] for m in indexed):
            ad.invalidate()

    def _end_step(self) -> None:
        super()._end_step()
        rm = self.managers["reward"]
        log = self._extras.get(self.extras_logging_key)
        if rm is not None and isinstance(log, LazyEpisodeLog) and log._snap is not None:
            rm._note_snapshot(log._snap)

    # -- reset ----------------------------------------------------------------------------------------
    def _reset_done(self, terminated: torch.Tensor, truncated: torch.Tensor) -> None:
        """managed_env.py:303-323 without the nonzero() sync when every reset can be expressed as a mask."""
        if type(self).reset is not ManagedEnvironment.reset:
            # a user subclass overrides reset(): honour it exactly like the reference does
            ids = self.done_ids(terminated, truncated, own=True)
            if ids.numel() > 0:
                self._done_ids = ids   # (reset() recognises THIS index list: the done envs, i.e. the termination masks)
                try:
                    self.reset(ids)
                finally:
                    self._done_ids = None
            return
        self._reset_with_mask(terminated, truncated, ids=None)

    def _reset_partition(self):
        """(managers whose reset is a section of gf_masked_reset, managers that need ``reset(ids)`` with an index list).
        Cached per configuration epoch: everything it depends on (manager classes, ``enabled``, on_reset entries and their params)
        drops the recorded step — and with it this cache — when it changes."""
        hit = self._partition_cache
        if hit is not None and hit[0] == self._trace_epoch:
            return hit[1], hit[2]
        fused, indexed = [], []
        for m in self._all_managers():
            if not _most_derived_reset_is_ours(m):
                indexed.append(m)
            elif m._fused_reset and getattr(m, "_can_fuse_reset", lambda: True)():
                fused.append(m)
            elif type(m).reset is BaseManager.reset:
                pass  # no-op reset (termination / observation managers)
            else:
                indexed.append(m)
        self._partition_cache = (self._trace_epoch, fused, indexed)
        return fused, indexed

    def _reset_with_mask(self, mask: torch.Tensor, mask2: Optional[torch.Tensor], ids) -> None:
        fused, indexed = self._reset_partition()
        a = self._reset_args  # persistent descriptor: a recorded step replays it in place
        C.memset(C.byref(a), 0, C.sizeof(a))
        ad = self._adapter
        if ad is not None:
            ad.begin_reset()   # the fused sections below register how their rows reach the simulator (SceneAdapter.on_push)
        a.mask = mask.data_ptr()
        a.mask2 = None if mask2 is None else mask2.data_ptr()
        self._fill_env_reset(a)
        for m in fused:
            m._fill_reset(a)
        a.stats = self.stats.ptr
        self._keep_reset = (mask, mask2)
        self.backend.call("masked_reset", a, owner=self)
        for m in fused:
            m._after_fused_reset(mask, mask2)
        pushes = ad is not None and bool(ad._pushes)
        if pushes:   # Genesis-shaped scene: the reset rows reach the simulator through its envs_idx setters
            if ids is None:
                ad.push_done(mask, mask2)
            elif not isinstance(ids, torch.Tensor) or ids.numel() > 0:
                ad.push(torch.as_tensor(ids, device=gs.device, dtype=torch.long))
        if indexed:
            rec = self.backend.tracer
            if ids is None and rec is not None and rec.part is None and not rec.tail_python:
                # the in-step reset while the step is being recorded: user code between native phases (see _manager_step) — the replay
                # finds the done envs of ITS step from the termination masks and calls the same managers
                from .managers._program import call_untraced
                fn = lambda self=self, indexed=list(indexed), mask=mask, mask2=mask2: self._indexed_reset(indexed, mask, mask2)
                rec.python(fn)
                call_untraced(self, fn)
            else:
                if ids is None:
                    ids = self.done_ids(mask, mask2, own=True)  # host sync: managers that need index lists
                if not isinstance(ids, torch.Tensor) or ids.numel() > 0:
                    for m in indexed:
                        m.reset(ids)
                    self._after_indexed_reset(indexed)
        self.invalidate_views()

    def _verify_adapter_setters(self) -> None:
        """Before the first full reset on a scene with Genesis' public surface only: does the simulator behave as the masked reset
        assumes (SceneAdapter.verify_setters)?  If not, the scene-side reset sections are not fused: every manager with one resets
        by index list, as in the reference (mdp/reset.py:102-124, position_action_manager.py:455-464)."""
        import warnings

        ad = self._adapter
        am = self.managers["action"]
        robot = getattr(self, "robot", None)
        if robot is None or not all(hasattr(robot, k) for k in ("set_pos", "set_quat", "get_pos", "get_quat")):
            ad.setters_verified = True   # nothing the masked reset would write through
            return
        ok = ad.verify_setters(robot, getattr(am, "dofs_idx", None) if am is not None else None)
        self._partition_cache = None
        if not ok:
            warnings.warn("genesis_forge_amd: the simulator's envs_idx setters do not read back what the masked reset assumes ("
                          + "; ".join(ad.setter_report) + ") - scene-side resets fall back to the index-list path", RuntimeWarning)

    def reset(self, env_ids: list[int] | None = None):
        """Reset one or more environments and every registered manager (managed_env.py:336-371)."""
        outside = not self._in_step
        if not outside and env_ids is not None and env_ids is self._done_ids:
            # the in-step reset of the done envs, reached through a user override of reset(): the index list IS the termination masks,
            # so the launches are the mask path's (persistent descriptors: a recorded step replays them with one native call)
            tr = self._tail_trace
            if tr is None or not tr.run_tail_segment("reset"):
                tm = self.managers["termination"]
                rec = self.backend.tracer
                if rec is not None:
                    rec.part = "reset"
                try:
                    self._reset_with_mask(tm._terminated_buf, tm._truncated_buf, ids=env_ids)
                finally:
                    if rec is not None:
                        rec.part = None
            return None, self.extras
        if outside:
            # a reset the training script calls between steps refills the persistent reset / masked-resample descriptors with
            # ITS masks — the ones a recorded step replays in place: drop the recording (two ordinary steps, then recorded again)
            self.invalidate_trace()
            self.stats.clear(self.backend)
            if self._adapter is not None:
                self._adapter.invalidate()   # between steps the simulator may have been edited: read it afresh
                if env_ids is None and self._adapter.setters_verified is None:
                    self._verify_adapter_setters()
        mask = self._ids_to_mask(env_ids)
        ids = env_ids if env_ids is not None else torch.arange(self.num_envs, device=gs.device)
        self._reset_with_mask(mask, None, ids=ids)
        obs = None
        if env_ids is None:
            obs = self.get_observations()
        if outside:
            log = self._extras.get(self.extras_logging_key)
            if isinstance(log, LazyEpisodeLog):
                snap = self.stats.snapshot()
                log.attach(snap)
                if self.managers["reward"] is not None:
                    self.managers["reward"]._note_snapshot(snap)
        return obs, self.extras

    def get_observations(self) -> torch.Tensor:
        """managed_env.py:373-398"""
        if len(self.managers["observation"]) > 0:
            if "observations" in self.extras and "policy" in self.extras["observations"]:
                return self.extras["observations"]["policy"]
            if "observations" not in self.extras:
                self.extras["observations"] = _obs_dict()
            tr = self._tail_trace
            if tr is not None and tr.run_tail_segment("obs"):   # the Python tail of a recorded step: one native call
                return self.extras["observations"].get("policy")
            rec = self.backend.tracer if self._in_step else None
            if rec is not None:
                rec.part = "obs"
            try:
                return self._observe_all()
            finally:
                if rec is not None:
                    rec.part = None
        return super().get_observations()

    def _observe_one(self, m):
        obs = m.get_observations()
        self.extras["observations"][m.name] = obs
        return obs

    def _observe_all(self):
        policy_obs = None
        rec = self.backend.tracer if self._in_step else None
        # (user-defined ObservationManager classes observe LAST, recorded or not: their code then sits behind the step's last native
        #  launch instead of between two observation phases, and the fused launch keeps the library managers — fuzz seed 310)
        ours = lambda m: _most_derived_is_ours(m, "get_observations") and _most_derived_is_ours(m, "_perform_observation")
        for m in sorted(self.managers["observation"], key=lambda m: not ours(m)):
            if rec is not None and not ours(m):
                # a user-defined ObservationManager class: its get_observations() is user code at this place of the step (the recording
                # keeps the place and calls it again there; the launch it makes itself belongs to it)
                from .managers._program import call_untraced
                fn = partial(self._observe_one, m)
                rec.python(fn)
                obs = call_untraced(self, fn)
            else:
                obs = self._observe_one(m)
            if m.name == "policy":
                policy_obs = obs
        return policy_obs

    def _user_get_observations(self) -> None:
        """An env whose get_observations() is overridden: the whole call is user code behind the step's native phases."""
        self._step_obs = self.get_observations()


# -- annotation type of the reference (managed_env.py:20-28) ---------------------------------------------------------------------------
from typing import TypedDict  # noqa: E402


class ManagersDict(TypedDict):
    contact: list
    entity: list
    command: list
    terrain: list
    action: Any
    observation: list
    reward: Any
    termination: Any
