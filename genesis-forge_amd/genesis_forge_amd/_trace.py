"""
Recorded step: trace one ordinary ``ManagedEnvironment.step()`` and replay it with a single native call.

The ordinary path spends ≈ 200 µs of Python per step marshalling eight launches whose descriptors do
not change from one step to the next (all buffers are persistent); the kernels themselves take ≈ 55 µs
at 65 536 envs.  A trace records the ``(phase, descriptor)`` sequence of two consecutive ordinary steps,
checks that they are the same objects in the same order, and from then on a step is:

    patch the few per-step fields in place (action pointer, scene tick, Philox stream ids in the order the
    ordinary path draws them, observation ring slot, statistics ring slot)  →  ``gf_run_ops``  →
    the same Python-side bookkeeping the ordinary path does (extras keys, lazy log fillers).

Results are bit-identical to the ordinary path (tests/test_trace.py).  Anything the trace cannot see
invalidates it and the env falls back to the ordinary path until two clean steps have been recorded
again: a mutated weight/param/scale (ConfigItem dirty hooks), parity-mode draws, an external command
controller, ``enabled`` toggles, a re-seed.  Configs with Python-evaluated terms, user-overridden
manager methods, or a scene without static buffers are never traced.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

from . import _native as nat
from ._stats import LazyEpisodeLog, StatsSnapshot


class Recorder:
    """Backend hook: collects (fn, descriptor, owner) while an ordinary step runs."""

    def __init__(self):
        self.calls: list = []

    def record(self, fn, args, owner):
        self.calls.append((fn, args, owner))

    def signature(self):
        return [(fn, C.addressof(args), id(owner)) for fn, args, owner in self.calls]


class StepTrace:
    def __init__(self, env, calls: list):
        self.env = env
        self.backend = env.backend
        self.epoch = env._trace_epoch
        self.patches: list[Callable] = []
        self.afters: list[Callable] = []
        n = len(calls) + 2
        self.ops = (nat.GfOp * n)()
        self.keep = [c[1] for c in calls]
        stats = env.stats
        k = 0
        self.ops[k].phase, self.ops[k].args = nat.GF_OP_STATS_CLEAR, stats.ptr
        k += 1
        for fn, args, owner in calls:
            self.ops[k].phase = nat.PHASE_OF_FN[fn]
            self.ops[k].args = C.addressof(args)
            k += 1
            self._hooks(fn, args, owner)
        self.use_native_copy = stats.group is None
        if self.use_native_copy:
            self.copy_args = nat.GfStatsCopyArgs()
            self.copy_args.src = stats.ptr
            self.ops[k].phase, self.ops[k].args = nat.GF_OP_STATS_COPY, C.addressof(self.copy_args)
            k += 1
            stats.ensure_native_events(self.backend)
        self.n_ops = k

    # -- per-phase hooks ----------------------------------------------------------------------------
    def _hooks(self, fn, args, owner):
        env = self.env
        if fn == "action_step":
            def patch(actions, a=args, owner=owner):
                a.actions_in = actions.data_ptr()
                owner._raw_actions = actions
            self.patches.append(patch)
            if not owner._quiet_action_errors:
                self.afters.append(owner._watch_flags)
        elif fn == "synth_scene_step":
            def patch(_actions, a=args, scene=owner):
                a.tick = scene.tick
                scene.tick += 1
            self.patches.append(patch)
        elif fn == "termination_step":
            self.afters.append(owner.manager._publish)
        elif fn == "reward_step":
            pass
        elif fn == "command_step":
            self.patches.append(owner._trace_patch(args))
        elif fn == "masked_reset":
            def patch(_actions, a=args, env=env):
                a.stream = env.next_stream()
                rm = env.managers["reward"]
                if rm is not None and rm.enabled and rm.logging_enabled:
                    rm._register_log()
            self.patches.append(patch)
            self.afters.append(env._after_masked_reset_traced)
        elif fn == "observe":
            self.patches.append(owner._trace_patch(args))
        elif fn == "contact_step":
            pass
        else:
            raise RuntimeError(f"untraceable phase {fn}")

    # -- replay ---------------------------------------------------------------------------------------
    def replay(self, actions):
        env = self.env
        env._begin_step_light()
        for p in self.patches:
            p(actions)
        snap = None
        if self.use_native_copy:
            snap = env.stats.native_slot(self.copy_args, self.backend)
        self.backend.run_ops(self.ops, self.n_ops)
        env._tick += 1  # scene advanced
        for f in self.afters:
            f()
        if snap is None:
            snap = env.stats.snapshot()
        env._finish_step_light(snap)
        tm, rm = env.managers["termination"], env.managers["reward"]
        obs = env.extras["observations"].get("policy") if len(env.managers["observation"]) > 0 else None
        return obs, rm._reward_buf if rm is not None else env._reward_buf, tm._terminated_buf, tm._truncated_buf, env.extras


def traceable(env) -> bool:
    """Static conditions under which an env's step may be recorded (see module docstring)."""
    from .managed_env import ManagedEnvironment, _most_derived_is_ours
    from .managers.action import PositionActionManager

    if type(env).step is not ManagedEnvironment.step or type(env).reset is not ManagedEnvironment.reset:
        return False
    if type(env).get_observations is not ManagedEnvironment.get_observations:
        return False
    if not getattr(env.scene, "gf_static_buffers", False):
        return False
    if env._draws:
        return False
    am, tm, rm = env.managers["action"], env.managers["termination"], env.managers["reward"]
    if am is None or tm is None or not isinstance(am, PositionActionManager) or not am.enabled or not tm.enabled:
        return False
    for m in env._all_managers() + env.managers["terrain"]:
        for meth in ("step", "reset", "get_observations", "_perform_observation", "handle_actions"):
            if hasattr(m, meth) and not _most_derived_is_ours(m, meth):
                return False
    if tm._dirty or len(tm._program.slots.exts) > 0:
        return False
    if rm is not None and (not rm.enabled or rm._dirty or len(rm._program.slots.exts) > 0):
        return False
    for cm in env.managers["command"]:
        if cm._external_controller is not None or not cm.enabled:
            return False
    for c in env.managers["contact"]:
        if not c.enabled:
            return False
    for om in env.managers["observation"]:
        if not om._traceable():
            return False
    for em in env.managers["entity"]:
        if not em.enabled or not em._can_fuse_reset():
            return False
    if not am._can_fuse_reset():
        return False
    return True
