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
controller, ``enabled`` toggles, a re-seed.  Configs with user-overridden manager methods or a scene
without static buffers are never traced.

Python-level terms (a lambda as a reward / termination term, an observation item no opcode covers) do not
stop a step from being recorded: the op list is cut in front of the op that consumes them, and the replay
runs ``gf_run_ops`` on the ops before the cut, then evaluates the callables — at exactly the point of the
step where the ordinary path (and the reference) calls them, on the same stream, no host sync — hands the
fresh columns to the descriptor, and continues with the next run of ops.  Callables of the termination and reward phases keep
the step on the fused post-physics launch: termination runs as a launch of its own, the callables run, one launch does
reward … observation with the termination masks as inputs (``GF_POST_TERMINATION_DONE``, ``_fuse_post``).

An env that overrides ``reset()`` is recorded up to the reset (``tail_python``): the user's ``reset()`` runs — by index list,
behind the ``nonzero()`` sync the reference pays too — and what ``super().reset(ids)`` and ``get_observations()`` launch is
replayed natively, one segment each, where the user's code reaches it (``tail_seg``; recorded from an ordinary step that had a
done env, or adopted from the first replayed step that resets one — walked phase by phase until then), with the launches pointed at the step's statistics slot.  A ``step()`` override
around ``super().step()`` is code of the training loop: the step inside is recorded as without it.  A recording also watches its descriptors: a phase call outside the replay that goes
through one of them (``reset([…])`` or ``resample_command([…])`` called by the training script between steps) drops it.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional

from . import _native as nat
from ._stats import LazyEpisodeLog, StatsSnapshot


class Recorder:
    """Backend hook: collects (fn, descriptor, owner) while an ordinary step runs."""

    def __init__(self, images: bool = False):
        self.calls: list = []
        #: Genesis-shaped scene: byte image of every descriptor at the moment it was launched, by address — two consecutive steps'
        #: images tell which pointer fields change from tick to tick (StepTrace._init_scene)
        self.images: Optional[dict] = {} if images else None
        self.tail_python = False
        #: launches of the Python tail, by part ("reset": inside the user's reset() → ManagedEnvironment.reset of the done envs;
        #: "obs": get_observations()) — a recorded step replays each part with one native call when it was seen (StepTrace.tail_seg)
        self.tail: dict = {}
        self.part = None

    def record(self, fn, args, owner):
        if self.images is not None:
            self.images[C.addressof(args)] = C.string_at(C.addressof(args), C.sizeof(args))
        if not self.tail_python:
            self.calls.append((fn, args, owner))
        elif self.part is not None:
            self.tail.setdefault(self.part, []).append((fn, args, owner))

    def python(self, fn) -> None:
        """A piece of user code the step runs between native phases (the ``step()`` / ``reset(ids)`` of a user-defined manager
        class): the recorded step calls ``fn`` again at this very point — after the launches recorded so far, before the next."""
        if not self.tail_python:
            self.calls.append(("__python__", fn, None))
        elif self.part is not None:
            self.tail.setdefault(self.part, []).append(("__python__", fn, None))

    def cut_tail(self):
        """Everything the step does from here on stays Python in the recorded step (an env that overrides ``reset()``: the
        index-list reset with its host sync, then the observations, run phase by phase exactly as in an ordinary step)."""
        self.tail_python = True

    def signature(self):
        return [(fn, C.addressof(args), id(owner)) if fn != "__python__" else (fn,) for fn, args, owner in self.calls] + ([("tail",)] if self.tail_python else [])


class Untraceable(Exception):
    """The step cannot be recorded (the reason is kept on the env: ``env._untraceable``)."""


class StepTrace:
    def __init__(self, env, calls: list, tail_python: bool = False, tail_calls: Optional[dict] = None, images: Optional[tuple] = None):
        self.env = env
        #: a scene with Genesis' public surface only (fresh getter tensors, envs_idx setters): the scene step and the state fetch
        #: run in Python in the middle of the replay, the snapshot's addresses reach the descriptors as call parameters
        self.adapter = env._adapter
        self.scene_plan: list = []
        self._scene_shapes: list = []     # the plan's tensor shapes in the recorded steps (checked every replayed tick)
        self._null_stub = None
        if env._adapter is not None:
            import torch
            from . import gs
            self._null_stub = torch.zeros(16, dtype=torch.int32, device=gs.device)
        self._images = images
        from .managed_env import ManagedEnvironment
        env_obs = type(env).get_observations is not ManagedEnvironment.get_observations
        self._env_obs_python = env_obs and not tail_python
        self._env_obs_tail = env_obs and tail_python   # (with a reset() override too: the Python tail calls it, the step returns ITS value)
        self.tail_python = tail_python
        self.tail_seg: dict = {}   # "reset" / "obs" → native segment of the Python tail (see _build_tail_segment)
        self.backend = env.backend
        self.epoch = env._trace_epoch
        #: user code between native phases (Recorder.python): (index of the recorded call it precedes, callable) — replayed as splits
        marks = []
        plain = []
        for c in calls:
            if c[0] == "__python__":
                marks.append((len(plain), c[1]))
            else:
                plain.append(c)
        calls = plain
        self.py_marks = marks
        self._term_call = next((c for c in calls if c[0] == "termination_step"), None)   # (its masks: inputs of a fused tail observation launch)
        self._tail_refs = None
        self.patches: list[Callable] = []   # Python-side per-step work that has Python semantics (live ranges, log registration)
        self.native: list = []              # GfReplayPatch entries: every per-step descriptor field, applied by gf_replay_step
        self.native_op: list = []           # … and the index of the op each entry belongs to (-1: none, applied first)
        self.afters: list = []   # (index of the op it follows, callable)
        self.splits: list = []   # (index of the op it precedes, callable): Python that must run in the middle of the step
        self._cur_op = 0
        n = len(calls) + 2
        self.ops = (nat.GfOp * n)()
        self.keep = [c[1] for c in calls]
        stats = env.stats
        k = 0
        self.ops[k].phase, self.ops[k].args = nat.GF_OP_STATS_CLEAR, stats.ptr
        k += 1
        self._gait_swaps: list = []
        self._late: set = set()
        first_post0 = self._post_start(calls)
        # (user code in the middle of the post-physics phases — a user manager's step() between reward and reset — keeps them off
        # the single fused launch: the phases on either side of it run as phase chains, gf_run_ops)
        marks_in_post = any(first_post0 < at < len(calls) for at, _ in marks)   # (user code BEHIND the last launch leaves the fused launch alone)
        self.post_refs = self._fuse_post(calls) if env.fuse_post_physics and not marks_in_post else None
        if self.post_refs is None:
            self._gait_swaps = []
            self._late = set()
        first_post = self._post_start(calls) if self.post_refs is not None else len(calls)
        self.post_split = self.post_refs is not None and bool(self.post_refs.flags & nat.GF_POST_TERMINATION_DONE)
        mark_i = 0
        post_index = -1   # final index of the fused launch's op
        scene_pre_at = None
        for idx, (fn, args, owner) in enumerate(calls):
            k_before = k
            while mark_i < len(marks) and marks[mark_i][0] <= idx:   # user code that ran before this call: a split in front of its op
                assert idx <= first_post or self.post_refs is None
                self.splits.append((k - 1, marks[mark_i][1]))
                mark_i += 1
                if scene_pre_at is not None:   # right behind the step() of a user-defined action manager class (the FIRST mark: it sends
                    self.splits.append((scene_pre_at, self._scene_pre))   # the targets itself) and in front of any other user code —
                    scene_pre_at = None                                    # a user termination manager's step() reads this tick's state
            if scene_pre_at is not None:
                self.splits.append((scene_pre_at, self._scene_pre))
                scene_pre_at = None
            if idx < first_post:
                self.ops[k].phase = nat.PHASE_OF_FN[fn]
                self.ops[k].args = C.addressof(args)
                k += 1
            elif idx == first_post and self.post_split:   # termination as a launch of its own, the fused launch behind it (Python between)
                self.ops[k].phase = nat.PHASE_OF_FN[fn]
                self.ops[k].args = C.addressof(args)
                k += 1
            elif idx == first_post + (1 if self.post_split else 0):
                self.ops[k].phase = nat.GF_OP_POST_PHYSICS
                self.ops[k].args = C.addressof(self.post_refs)
                k += 1
                post_index = k - 2
            elif fn == "history_unroll" or idx in self._late:   # the gather of a ring-kept history — and an observation manager with a
                self.ops[k].phase = nat.PHASE_OF_FN[fn]         # Python-level item — follow the fused launch as ops of their own
                self.ops[k].args = C.addressof(args)
                k += 1
            # index of this call's op once the leading STATS_CLEAR op is dropped (below).  A call that is PART of the fused launch has
            # no op of its own: its per-step fields belong to the fused launch's op — not to whatever op was assigned last (a late
            # observation manager's launch can sit between two fused calls in the recorded order, and with per-piece patch tables its
            # output rotation would then be applied one piece too late: fuzz seeds 57 / 115 / 132 / 139)
            self._cur_op = k - 2 if (k != k_before or post_index < 0) else post_index
            self._hooks(fn, args, owner)
            self.native_op.extend([self._cur_op] * (len(self.native) - len(self.native_op)))
            if fn == "action_step" and self.adapter is not None:
                if owner is not None:
                    self.splits.append((self._cur_op + 1, self._scene_pre))   # control_dofs_position → scene.step() → state fetch
                else:
                    scene_pre_at = self._cur_op + 1
            pre = owner._trace_pre(args) if hasattr(owner, "_trace_pre") else None
            if pre is not None:
                assert idx < first_post + (2 if self.post_split else 0) or idx in self._late, "a phase with Python-level terms cannot be part of the fused launch"
                self.splits.append((self._cur_op, pre))
        for _at, f in marks[mark_i:]:   # user code behind the last launch
            self.splits.append((k - 1, f))
            if scene_pre_at is not None:
                self.splits.append((scene_pre_at, self._scene_pre))
                scene_pre_at = None
        if scene_pre_at is not None:
            self.splits.append((scene_pre_at, self._scene_pre))
        self.native.extend(self._gait_swaps)   # after the gait managers' own patches (those refill the descriptors)
        post_at = next((i - 1 for i in range(k) if self.ops[i].phase == nat.GF_OP_POST_PHYSICS), -1)
        self.native_op.extend([post_at] * (len(self.native) - len(self.native_op)))   # (they serve the fused launch)
        # single process: statistics go to a device ring slot per step (no memset, no copy); with a process group the
        # per-step all-reduce path is kept (clear op here, packed + reduced + copied by StepStats.snapshot)
        self.use_ring = stats.group is None
        self.stat_fields = [c[1] for c in calls if hasattr(c[1], "stats") and c[1].stats]
        self.action_args = next(c[1] for c in calls if c[0] == "action_step")
        for i in range(k - 1):  # drop the leading STATS_CLEAR op: ring slots are zeroed by the previous step's action kernel
            self.ops[i].phase, self.ops[i].args = self.ops[i + 1].phase, self.ops[i + 1].args
        k -= 1
        #: process group + batched reduction: the previous step's statistics are folded by this step's action kernel (as in the
        #: single-process ring) instead of a pack launch per step; rows are all-reduced K at a time (StepStats.vec_ring_reduce)
        self.fold_mode = (not self.use_ring) and stats.reduce_every > 1
        if self.use_ring:
            stats.ensure_ring()
        else:
            stats.ensure_vec_ring()
            if not self.fold_mode:
                self.pack_args = nat.GfStatsPackArgs()
                self.ops[k].phase, self.ops[k].args = nat.GF_OP_STATS_PACK, C.addressof(self.pack_args)
                k += 1
        self.n_ops = k
        # the statistics ring slots of a step arrive as call parameters (cur, next-to-zero, previous, its vector row, last_reset);
        # behind them, on a Genesis-shaped scene, the addresses of this tick's state tensors (_init_scene)
        self.params = (C.c_void_p * 5)()
        self.n_params = 5
        if self.use_ring or self.fold_mode:
            P = nat.GfReplayPatch
            for a in self.stat_fields:
                self.native.append(P(nat.GF_PATCH_PARAM, 0, nat.field_addr(a, "stats"), None, None))
            aa = self.action_args
            for idx, name in ((1, "stats_zero"), (2, "stats_fold_src"), (3, "stats_fold_dst"), (4, "stats_last_reset")):
                self.native.append(P(nat.GF_PATCH_PARAM, idx, nat.field_addr(aa, name), None, None))
            self._last_reset_ptr = stats.last_reset.data_ptr() if self.use_ring else None   # (group ring: gf_stats_last_reset)
        # gf_replay_step: the whole table, then the ops, in ONE native call (the patch-only variant serves steps whose ops are
        # replayed in pieces around Python-level terms, or as a hipGraph)
        self.native_op.extend([-1] * (len(self.native) - len(self.native_op)))   # statistics slots: call parameters, no order
        self.patch_table = (nat.GfReplayPatch * max(1, len(self.native)))(*self.native)
        self.replay_desc = nat.GfReplay(C.addressof(self.ops), k, len(self.native), C.addressof(self.patch_table), C.addressof(env._rng_c))
        self.patch_desc = nat.GfReplay(None, 0, len(self.native), C.addressof(self.patch_table), C.addressof(env._rng_c))
        #: the descriptors this recording froze: a phase call that goes through one of them from now on (a manager method the
        #: training script calls between steps) makes the recording stale (Backend._note_call, fresh())
        self.arg_set = {C.addressof(c[1]) for c in calls}
        if self.adapter is not None:
            self._init_scene([c[1] for c in calls])
        # (both parts or none: the observation descriptors of a step WITH a reset carry the stale-quaternion stash and the termination
        # masks — with all-false masks they also describe a step without one; the descriptors of a step without a reset do not)
        self._tail_tries = 0
        if tail_python:
            self._adopt_tail(tail_calls)
        b = self.backend
        b.__dict__.setdefault("dirty", set()).difference_update(self.arg_set)
        b.__dict__.setdefault("watched", set()).update(self.arg_set)
        #: hipGraph of this step's launches (built by the library on first replay; HIP backend only)
        self.graph = C.c_void_p() if hasattr(self.backend, "run_ops_graph") and not self.splits and not tail_python else None
        #: the op list cut at the splits: (first op, count, callable to run before it or None)
        self.segments = []
        if self.splits:
            cuts = [0] + [i for i, _ in self.splits] + [k]
            pres = [None] + [f for _, f in self.splits]
            for (a0, a1), pre in zip(zip(cuts[:-1], cuts[1:]), pres):
                sub = (nat.GfOp * (a1 - a0)).from_buffer(self.ops, a0 * C.sizeof(nat.GfOp)) if a1 > a0 else None
                self.segments.append((a0, a1 - a0, sub, pre))
            # Each piece of the op list goes out with its patches (one native call per piece).  The table stays in CALL order — stream
            # ids are handed out in the order the ordinary step draws them — and a piece applies the not yet applied PREFIX of it up
            # to the last entry one of its own ops needs: user code that runs between two pieces may draw Philox streams itself (a
            # user manager's step() calling the base class' resample), so the entries of the calls behind it must not run before it;
            # but an observation manager that is part of the fused launch can sit BEHIND a late manager's launch in call order, and
            # then the late manager's entries go out early with it (fuzz seeds 57 / 115 / 132: its stream id comes first).  The
            # statistics slots (call parameters, no order) go with the first piece; on a Genesis-shaped scene the piece behind the
            # scene split also carries every descriptor's snapshot pointers.
            scene_tab = list(getattr(self, "_scene_tab", []))
            ordered = [i for i, at in enumerate(self.native_op) if at >= 0]
            unordered = [self.native[i] for i, at in enumerate(self.native_op) if at < 0]
            done_upto = 0
            segs = []
            for j, (a0, cnt, sub, pre) in enumerate(self.segments):
                mine = list(scene_tab) if (self.adapter is not None and pre == self._scene_pre) else []
                if j == 0:
                    mine += unordered
                need = [pos for pos, i in enumerate(ordered) if a0 <= self.native_op[i] < a0 + cnt]
                end = len(ordered) if j == len(self.segments) - 1 else max([done_upto] + [pos + 1 for pos in need])
                mine += [self.native[i] for i in ordered[done_upto:end]]
                done_upto = max(done_upto, end)
                table = (nat.GfReplayPatch * max(1, len(mine)))(*mine)
                desc = nat.GfReplay(C.addressof(self.ops) + a0 * C.sizeof(nat.GfOp) if cnt else None, cnt, len(mine), C.addressof(table), C.addressof(env._rng_c))
                segs.append((a0, cnt, pre, desc, table))
            self.segments = segs
            assert sum(d.num_patches for *_x, d, _t in segs) == len(self.native) + len(scene_tab)

    def fresh(self) -> bool:
        """No descriptor of this recording has been used by a phase call outside its replay since it was made."""
        d = self.backend.__dict__.get("dirty")
        return not d or d.isdisjoint(self.arg_set)

    def __del__(self):
        b, mine = getattr(self, "backend", None), getattr(self, "arg_set", None)
        if b is not None and mine:
            for name in ("watched", "dirty"):
                st = b.__dict__.get(name)
                if st:
                    st.difference_update(mine)
        g = getattr(self, "graph", None)
        if g is not None and g.value:
            try:
                self.backend.graph_destroy(g)
            except Exception:
                pass

    # -- a scene with Genesis' public surface only --------------------------------------------------------------------------
    def _scene_patches(self, descs: list) -> list:
        """GF_PATCH_PARAM_OFFSET entries for every pointer field of ``descs`` that addresses a tensor of this tick's snapshot.
        The fetch plan only ever grows (a tail adopted later may read state the main part does not): parameter indices stay."""
        ad = self.adapter
        plan = ad.plan()
        if [k for k, _ in plan[:len(self.scene_plan)]] != [k for k, _ in self.scene_plan]:
            raise Untraceable("the scene snapshot of this tick does not extend the recorded fetch plan")
        if len(plan) > len(self.scene_plan):
            self.scene_plan = plan
            self._scene_shapes = [tuple(ad.peek(key).shape) for key, _f in plan]
            params = (C.c_void_p * (5 + len(plan)))()
            for i in range(self.n_params):
                params[i] = self.params[i]
            for i, (key, _f) in enumerate(plan):
                params[5 + i] = ad.peek(key).data_ptr()
            self.params, self.n_params = params, 5 + len(plan)
        skip = set()
        for p in self.native:
            skip.update(t for t in (p.target, p.target2) if t)
        patches, covered = ad.attribute(descs, self.scene_plan, skip)
        self._scene_covered = getattr(self, "_scene_covered", set()) | covered | skip
        P = nat.GfReplayPatch
        return [P(nat.GF_PATCH_PARAM_OFFSET, 5 + i, addr, None, off if off else None) for addr, i, off in patches]

    def _init_scene(self, descs: list) -> None:
        from ._scene_adapter import PER_STEP_FIELDS, changed_pointer_fields
        tab = self._scene_patches(descs)
        if not tab:
            raise Untraceable("no descriptor reads the scene snapshot")
        # every pointer field that differed between the two recorded steps must be explained: a snapshot tensor (patched above),
        # a field another patch writes, or one of the per-step fields the step's own bookkeeping sets
        if self._images is not None and self._images[0] is not None:
            before, now = self._images
            for addr, name in changed_pointer_fields(descs, [before.get(C.addressof(d)) for d in descs], [now.get(C.addressof(d)) for d in descs]):
                leaf = name.rsplit(".", 1)[-1]
                if addr in self._scene_covered or leaf in PER_STEP_FIELDS or leaf.startswith("ext["):
                    continue
                raise Untraceable(f"{name} changes from step to step and is neither scene state nor a per-step field")
        self._scene_tab = tab
        self.scene_table = (nat.GfReplayPatch * len(tab))(*tab)

    def _scene_pre(self) -> None:
        """What the ordinary step does between the action phase and the first post-physics phase (managed_env.py:290-292,
        position_action_manager.py:417), then the snapshot: each getter of the plan once, the new addresses into the descriptors."""
        env, am = self.env, self.action_owner
        if am is not None:   # (a user-defined action manager class has sent its targets itself: its step() ran just before)
            env.robot.control_dofs_position(am._actions, am.dofs_idx)
        env.scene.step()
        pr = self.params
        shapes = self._scene_shapes
        for i, t in enumerate(self.adapter.refetch(self.scene_plan)):
            # (applied by the piece of the op list that follows: its table carries the pointer patches.  An EMPTY tensor — a tick
            #  without a single contact — has no address; no kernel reads through it either: any non-null value will do)
            pr[5 + i] = t.data_ptr() or self._null_stub.data_ptr()
            if i < len(shapes) and tuple(t.shape) != shapes[i]:
                self._scene_shape_changed(i, t)

    def _scene_shape_changed(self, i: int, t) -> None:
        """A tensor of the fetch plan came back with another shape than in the recorded steps.  Genesis pads the collider's contact
        arrays to the current tick's contact count (contact_manager.py:391-426 takes its sizes from the fresh tensors every step), so
        for those the scalar fields of the recorded ContactManager descriptors follow; anything else is not a step this recording
        describes."""
        key = self.scene_plan[i][0]
        old = self._scene_shapes[i]
        if key and key[0] == "contacts" and t.dim() == len(old) and tuple(t.shape)[:1] == old[:1]:
            c = int(t.shape[1]) if t.dim() > 1 else 0
            for a in getattr(self, "_contact_args", []):
                a.num_contacts = c
            self._scene_shapes[i] = tuple(t.shape)
            return
        raise RuntimeError(f"recorded step: the scene's {key} changed shape from {old} to {tuple(t.shape)} between ticks; "
                           "call env.invalidate_trace() after changing the scene")

    # -- the Python tail of an env that overrides reset(), part by part ----------------------------------------------------
    def _fuse_tail_obs(self, calls, reset_args):
        """The observation launches of the tail as ONE launch of the fused kernel's observation waves (GF_POST_OBSERVE_ONLY: the
        masks are inputs, nothing is reset) when the library takes the combination — up to two managers, their gathers behind.
        Returns the replacement call list, or ``calls`` unchanged."""
        term = self._term_call
        obs = [c for c in calls if c[0] == "observe"]
        if term is None or reset_args is None or not 1 <= len(obs) <= nat.GF_POST_MAX_OBS:
            return calls
        if self.adapter is not None:
            # (a scene whose getters return new tensors every tick: the reset descriptor's scene pointers are refreshed only in steps
            #  that reset an env, so in the others they would not match this tick's — the observation launches stay their own)
            return calls
        if os.environ.get("GF_NO_TAIL_FUSE", "0") == "1":
            return calls
        refs = nat.GfPostRefs()
        refs.flags = nat.GF_POST_OBSERVE_ONLY
        refs.termination, refs.reset = C.addressof(term[1]), C.addressof(reset_args)
        refs.num_observe = len(obs)
        for m, o in enumerate(obs):
            refs.observe[m] = C.addressof(o[1])
        if not self.backend.post_check(refs):
            return calls
        self._tail_refs = refs   # (kept alive with the recording)
        first = next(k for k, c in enumerate(calls) if c[0] == "observe")
        rest = [c for c in calls if c[0] != "observe"]
        return rest[:first] + [("post_obs", refs, obs)] + rest[first:]

    def _build_tail_segment(self, calls, reset_args=None):
        """The launches one part of the Python tail made in the ordinary step (the in-step reset of the done envs by the
        termination masks; the observations) as a patch table + op list of their own.  The user's reset() still runs — its code
        before and after ``super().reset(ids)`` sees exactly what it sees in an ordinary step — but what ``super().reset(ids)`` and
        ``get_observations()`` do is one native call each instead of a Python walk over every manager."""
        if not calls or any(hasattr(o, "_trace_pre") and o._trace_pre(a) is not None for _, a, o in calls):
            return None   # (a Python-level observation item: that part stays phase by phase)
        if any(fn not in nat.PHASE_OF_FN for fn, _, _ in calls):
            return None
        # gathers behind all observation launches: two of them then share a launch (gf_run_ops)
        calls = [c for c in calls if c[0] != "history_unroll"] + [c for c in calls if c[0] == "history_unroll"]
        described = [c[1] for c in calls]   # every descriptor a launch reads, fused or not
        if reset_args is not None:
            calls = self._fuse_tail_obs(calls, reset_args)
        saved = (self.native, self.patches, self.afters, self._cur_op)
        self.native, self.patches, self.afters = [], [], []
        try:
            ops = (nat.GfOp * len(calls))()
            P = nat.GfReplayPatch
            for k, (fn, args, owner) in enumerate(calls):
                self._cur_op = k
                if fn == "post_obs":   # the fused observation launch: the hooks and patches are the member launches' own
                    ops[k].phase, ops[k].args = nat.GF_OP_POST_PHYSICS, C.addressof(args)
                    members = owner
                else:
                    ops[k].phase, ops[k].args = nat.PHASE_OF_FN[fn], C.addressof(args)
                    members = [(fn, args, owner)]
                for mfn, margs, mowner in members:
                    self._hooks(mfn, margs, mowner)
                    if hasattr(margs, "stats") and margs.stats:
                        self.native.append(P(nat.GF_PATCH_PARAM, 0, nat.field_addr(margs, "stats"), None, None))
            if self.adapter is not None:
                self.native.extend(self._scene_patches(described))
            table = (nat.GfReplayPatch * max(1, len(self.native)))(*self.native)
            desc = nat.GfReplay(C.addressof(ops), len(calls), len(self.native), C.addressof(table), C.addressof(self.env._rng_c))
            return {"ops": ops, "table": table, "desc": desc, "patches": self.patches, "afters": [f for _, f in self.afters],
                    "keep": [c[1] for c in calls] + described, "fused_obs": any(c[0] == "post_obs" for c in calls)}
        finally:
            self.native, self.patches, self.afters, self._cur_op = saved

    def _tail_native_ok(self) -> bool:
        """The native tail replays only what went through ``backend.call``: every manager's reset must be a section of the masked
        reset (or a no-op) — a manager that needs ``reset(ids)`` (a Python ``on_reset`` entry, a user manager class, a user
        ``resample_command``) is walked phase by phase — and the observation descriptors must be ones a recording may freeze."""
        env = self.env
        _, indexed = env._reset_partition()
        if indexed:
            return False
        if any(not om._traceable() for om in env.managers["observation"]):
            return False
        if any(not em.enabled for em in env.managers["entity"]):
            return False
        return True

    def _adopt_tail(self, tail_calls) -> None:
        if not tail_calls or not tail_calls.get("reset") or not tail_calls.get("obs"):
            return
        if not self._tail_native_ok():
            self._tail_tries = 1 << 30   # the tail stays a Python walk for the life of this recording
            return
        try:
            reset_args = next((c[1] for c in tail_calls["reset"] if c[0] == "masked_reset"), None)
            segs = {"reset": self._build_tail_segment(tail_calls["reset"]),
                    "obs": self._build_tail_segment(tail_calls["obs"], reset_args)}
        except Untraceable:
            segs = {"reset": None}
        if any(v is None for v in segs.values()):
            self._tail_tries = 1 << 30   # a part that cannot be replayed natively (a Python-level observation item): stop trying
            return
        self.tail_seg = segs
        if segs["obs"].get("fused_obs"):
            from . import _programs

            _programs.on_recorded(self.env, self._tail_refs)   # (a structure no built-in program matches gets its own: GF_JIT)
        mine = {C.addressof(c[1]) for part in ("reset", "obs") for c in tail_calls[part]}
        self.arg_set.update(mine)
        b = self.backend
        b.__dict__.setdefault("dirty", set()).difference_update(mine)
        b.__dict__.setdefault("watched", set()).update(mine)

    def run_tail_segment(self, part: str) -> bool:
        """Replay one part of the Python tail natively; False when that part was not recorded (the caller walks the managers)."""
        seg = self.tail_seg.get(part)
        if seg is None:
            return False
        if self.env._soft_dirty:
            # the user's reset() override assigned a weight / param / scale / noise (a curriculum): the part of the tail behind it reads
            # the new number in this very step, as the ordinary step does — refreshed in place, or the recording goes (then: below)
            self.env._refresh_soft()
        if self.epoch != self.env._trace_epoch or not self.fresh():
            # something the frozen descriptors depend on was mutated since this step began (the user's reset() override ran a
            # curriculum): the rest of the tail walks the managers, which read the live values like the ordinary step does
            return False
        for p in seg["patches"]:
            p(None)
        self.backend.replay_step(seg["desc"], None, self.params, self.n_params)
        for f in seg["afters"]:
            f()
        return True

    # -- fused post-physics launch -----------------------------------------------------------------------
    @staticmethod
    def _post_start(calls) -> int:
        for i, (fn, _, _) in enumerate(calls):
            if fn == "termination_step":
                return i
        return len(calls)

    def _fuse_post(self, calls):
        """If the tail of the step is [termination, reward?, command.step*, reset, command.reset*, observe*], describe it
        to gf_post_physics_step (one launch) — provided the library agrees the combination is fusable."""
        i = self._post_start(calls)
        tail = calls[i:]
        fns = [c[0] for c in tail]
        if not fns or fns[0] != "termination_step":
            return None
        refs = nat.GfPostRefs()
        py = [i for i, c in enumerate(tail) if hasattr(c[2], "_trace_pre") and c[2]._trace_pre(c[1]) is not None]
        if py:
            # Python runs between these phases.  Callables of the termination and reward phases (the common customisation: a lambda as
            # a reward term) still leave everything behind the termination phase fusable: termination runs as a launch of its own,
            # the callables run where the reference calls them (before their phase, after the earlier ones: a reward callable may
            # read this step's termination buffers), and ONE launch does reward … observation with the termination masks as inputs
            # (GF_POST_TERMINATION_DONE).  A Python-level OBSERVATION item sees the post-reset state: the manager that owns it is left
            # out of the fused launch and observes behind it, after its callables, as a launch of its own (self._late).
            if any(fns[i] not in ("termination_step", "reward_step", "observe") for i in py) or len(fns) < 2:
                return None
            if any(fns[i] in ("termination_step", "reward_step") for i in py):
                refs.flags = nat.GF_POST_TERMINATION_DONE
        late_obs = {k for k in py if fns[k] == "observe"}
        self._late = set()
        refs.termination = C.addressof(tail[0][1])
        j = 1
        if j < len(fns) and fns[j] == "reward_step":
            refs.reward = C.addressof(tail[j][1])
            j += 1
        steps, gsteps = [], []
        while j < len(fns) and fns[j] in ("command_step", "gait_step") and tail[j][1].mode == nat.GF_CMD_STEP:
            (steps if fns[j] == "command_step" else gsteps).append(tail[j])
            j += 1
        if j == len(fns) and self.tail_python and not py and (steps or gsteps) and os.environ.get("GF_NO_TAIL_FUSE", "0") != "1":
            # An env whose reset() is overridden: the recording ends in front of the reset (the user's reset(ids) and the observations
            # are the Python tail).  Termination, rewards and the command / gait steps still make ONE launch — GF_POST_NO_RESET: the
            # masks are written, no env is treated as done — instead of a one-wave phase chain.
            if len(steps) > nat.GF_POST_MAX_CMD or len(gsteps) > nat.GF_POST_MAX_GAIT:
                return None
            refs.flags = nat.GF_POST_NO_RESET
            refs.num_command, refs.num_gait = len(steps), len(gsteps)
            for c, st in enumerate(steps):
                refs.command_step[c] = C.addressof(st[1])
            self._gait_swaps = []
            P = nat.GfReplayPatch
            reward_args = tail[1][1] if refs.reward else None
            for g, st in enumerate(gsteps):
                refs.gait_step[g] = C.addressof(st[1])
                mgr = st[2]
                refs.gait_flags_next[g] = mgr._wave_flags_next.data_ptr()
                # the launch reads the bytes the previous step left and writes its own into the manager's OTHER buffer, as the full
                # fused launch does; the masked gait launch of the tail then finds the manager's CURRENT buffer (the rotor has
                # advanced) and updates the reset blocks' bytes in place
                sw = [P(nat.GF_PATCH_ROTATE, 0, nat.field_addr(st[1], "wave_flags"), C.addressof(refs) + nat.GfPostRefs.gait_flags_next.offset + 8 * g,
                        C.addressof(mgr._flags_rotor))]
                if reward_args is not None and reward_args.gait_wave_flags:
                    sw.append(P(nat.GF_PATCH_COPY, 0, nat.field_addr(reward_args, "gait_wave_flags"), None, nat.field_addr(st[1], "wave_flags")))
                self._gait_swaps.extend(sw)
            return refs if self.backend.post_check(refs) else None
        if j >= len(fns) or fns[j] != "masked_reset":
            return None
        refs.reset = C.addressof(tail[j][1])
        reward_args = tail[1][1] if refs.reward else None
        j += 1
        resets, gresets = [], []
        while j < len(fns) and fns[j] in ("command_step", "gait_step") and tail[j][1].mode == nat.GF_CMD_MASKED:
            (resets if fns[j] == "command_step" else gresets).append(tail[j])
            j += 1
        obs = []
        while j < len(fns) and fns[j] in ("observe", "history_unroll"):   # (a manager that keeps its history as a ring: frame, then gather)
            if fns[j] == "observe":
                if j not in late_obs and len(obs) >= nat.GF_POST_MAX_OBS:
                    late_obs.add(j)           # more managers than the fused launch holds: the further ones observe behind it
                if j in late_obs:
                    self._late.add(i + j)     # (indices into `calls`) this manager's launch, and its gather, stay ops of their own
                else:
                    obs.append(tail[j])
            elif any(tail[k][2] is tail[j][2] for k in late_obs):
                self._late.add(i + j)
            j += 1
        if late_obs and j < len(fns) and fns[j] == "rollout_write":
            return None   # (a rollout row out of an observation that is not part of the fused launch: keep the chains)
        if j < len(fns) and fns[j] == "rollout_write":   # learner.RolloutStorage: its rows are stored by the same launch
            refs.rollout = C.addressof(tail[j][1])
            pol = next((m for m in self.env.managers["observation"] if m.name == tail[j][2].obs_name), None)
            if pol is not None and pol._unrolled:
                tail[j][1].obs_out = None   # … except the observation row of a ring-kept history: its gather writes it (learner.py)
            j += 1
        if j != len(fns) or len(steps) != len(resets) or len(steps) > nat.GF_POST_MAX_CMD or len(obs) > nat.GF_POST_MAX_OBS:
            return None
        if len(gsteps) != len(gresets) or len(gsteps) > nat.GF_POST_MAX_GAIT:
            return None
        for s, r in zip(steps + gsteps, resets + gresets):
            if s[2] is not r[2]:
                return None
        refs.num_gait = len(gsteps)
        self._gait_swaps = []
        for g, (s, r) in enumerate(zip(gsteps, gresets)):
            refs.gait_step[g] = C.addressof(s[1])
            refs.gait_reset[g] = C.addressof(r[1])
            mgr = s[2]
            refs.gait_flags_next[g] = mgr._wave_flags_next.data_ptr()
            # One launch reads the swing / stance bytes the previous step left and writes the bytes of the state it leaves into
            # the manager's OTHER buffer (GfPostRefs): rotate the manager's two buffers, the other descriptors follow
            P = nat.GfReplayPatch
            sw = [P(nat.GF_PATCH_ROTATE, 0, nat.field_addr(s[1], "wave_flags"), C.addressof(refs) + nat.GfPostRefs.gait_flags_next.offset + 8 * g,
                    C.addressof(mgr._flags_rotor)),
                  P(nat.GF_PATCH_COPY, 0, nat.field_addr(r[1], "wave_flags"), None, nat.field_addr(s[1], "wave_flags"))]
            if reward_args is not None and reward_args.gait_wave_flags:
                sw.append(P(nat.GF_PATCH_COPY, 0, nat.field_addr(reward_args, "gait_wave_flags"), None, nat.field_addr(s[1], "wave_flags")))
            self._gait_swaps.extend(sw)
        refs.num_command, refs.num_observe = len(steps), len(obs)
        for c, (s, r) in enumerate(zip(steps, resets)):
            refs.command_step[c] = C.addressof(s[1])
            refs.command_reset[c] = C.addressof(r[1])
        for m, o in enumerate(obs):
            refs.observe[m] = C.addressof(o[1])
        return refs if self.backend.post_check(refs) else None

    # -- per-phase hooks ----------------------------------------------------------------------------
    def _hooks(self, fn, args, owner):
        env = self.env
        P = nat.GfReplayPatch
        if fn == "action_step":
            self.action_owner = owner
            self.native.append(P(nat.GF_PATCH_ACTIONS, 0, nat.field_addr(args, "actions_in"), None, None))
            # (owner None: the env's bookkeeping launch in front of a user-defined action manager class, whose own step() is a python phase)
            if owner is not None and not owner._quiet_action_errors:
                self.afters.append((self._cur_op, owner._watch_flags))
        elif fn == "synth_scene_step":
            self.native.append(P(nat.GF_PATCH_COUNTER, 0, nat.field_addr(args, "tick"), None, C.addressof(owner._tick_c)))
        elif fn == "termination_step":
            self.afters.append((self._cur_op, owner.manager._publish))
        elif fn == "reward_step":
            pass
        elif fn in ("command_step", "gait_step"):
            self.patches.append(owner._trace_patch(args))
            self.native.extend(owner._trace_native(args))
        elif fn == "masked_reset":
            self.native.append(P(nat.GF_PATCH_STREAM, 0, nat.field_addr(args, "stream"), None, None))
            rm = env.managers["reward"]
            if rm is not None:
                def patch(_actions, rm=rm):
                    if rm.enabled and rm.logging_enabled:
                        rm._register_log()
                self.patches.append(patch)
            self.afters.append((self._cur_op, env._after_masked_reset_traced))
        elif fn == "observe":
            self.native.extend(owner._trace_native(args))
            fresh = owner._trace_fresh_patch(args)
            if fresh is not None:
                self.patches.append(fresh)
            if not owner._unrolled:
                self.afters.append((self._cur_op, owner._trace_after))
        elif fn == "history_unroll":
            patch, native = owner._trace_unroll(args)
            if patch is not None:
                self.patches.append(patch)
            self.native.extend(native)
            self.afters.append((self._cur_op, owner._trace_after))
        elif fn == "contact_step":
            # (an adapter scene re-checks the contact arrays' shapes every replayed tick: _scene_pre)
            if not hasattr(self, "_contact_args"):
                self._contact_args = []
            self._contact_args.append(args)
        elif fn == "rollout_write":
            pol = next(m for m in env.managers["observation"] if m.name == owner.obs_name)
            fused = self.post_refs is not None and bool(self.post_refs.rollout)
            self.patches.append(owner._trace_patch(args, pol if fused and pol._unrolled else None))
            self.native.extend(owner._trace_native(args, pol, fused))
        else:
            raise RuntimeError(f"untraceable phase {fn}")

    # -- replay ---------------------------------------------------------------------------------------
    def replay(self, actions):
        env = self.env
        env._begin_step_light()
        if self.action_owner is not None:
            self.action_owner._raw_actions = actions
        else:
            env._step_actions = actions
        for p in self.patches:
            p(actions)
        snap = None
        pr = self.params
        if self.use_ring:
            cur, nxt, prev, prev_vec, snap = env.stats.ring_next()
            pr[0], pr[1], pr[2], pr[3] = cur, nxt, prev, prev_vec
            pr[4] = self._last_reset_ptr if prev is not None else None
        elif self.fold_mode:
            slot, cur, nxt, prev, prev_vec, snap = env.stats.group_ring_next()
            pr[0], pr[1], pr[2], pr[3] = cur, nxt, prev, prev_vec
        else:
            slot, cur, nxt, vec = env.stats.vec_ring_next()
            self.pack_args.src, self.pack_args.dst = cur, vec
            for a in self.stat_fields:
                a.stats = cur
            self.action_args.stats_zero = nxt
        aptr = actions.data_ptr()
        done = 0        # afters already run
        ticked = False  # the scene op has been enqueued and the views cache invalidated for it
        if self.segments:
            for first, count, pre, desc, _table in self.segments:
                if pre is not None:
                    # the ordinary path has finished every earlier phase — launch AND Python bookkeeping — when it calls a
                    # Python-level term: views of the scene are stale after the scene op, earlier phases' hooks have run
                    if not ticked:
                        env._tick += 1
                        ticked = True
                    while done < len(self.afters) and self.afters[done][0] < first:
                        self.afters[done][1]()
                        done += 1
                    # (launches the user code makes itself — a user manager's `super().step()` — count into THIS step's statistics slot)
                    env.stats.ptr_override = cur
                    try:
                        pre()
                    finally:
                        env.stats.ptr_override = None
                    if env._soft_dirty:
                        # the user code assigned a weight / param value: a phase that runs BEHIND it reads the new number in this very
                        # step, as it does in the ordinary step (where the manager compiles at its phase)
                        env._refresh_soft()
                if count or desc.num_patches:
                    self.backend.replay_step(desc, aptr, pr, self.n_params)
        elif self.graph is not None and self.backend.graph_enabled:
            self.backend.replay_step(self.patch_desc, aptr, pr, self.n_params)
            self.backend.run_ops_graph(self.graph, self.ops, self.n_ops)
        else:
            self.backend.replay_step(self.replay_desc, aptr, pr, self.n_params)   # patch table + ops: the one native call of the step
        if self.fold_mode:
            env.stats.group_ring_after(slot)   # every K-th step: the single collective of the path, asynchronous, K rows
        elif not self.use_ring:
            snap = env.stats.vec_ring_reduce(slot)  # the single collective of the path, asynchronous
        if not ticked:
            env._tick += 1  # scene advanced
        for _, f in self.afters[done:]:
            f()
        tm, rm = env.managers["termination"], env.managers["reward"]
        obs_tail = None
        if self.tail_python:
            # reset (the user's override, by index list, behind the same nonzero() sync the reference pays) and observations,
            # phase by phase; their launches write their statistics into this step's ring slot
            env.stats.ptr_override, env._in_step, env._tail_trace = cur, True, self
            rec = None
            if not self.tail_seg and self._tail_tries < 64:
                # the ordinary steps this recording came from had no done env, so their tail could not be recorded (the observation
                # descriptors of a step with a reset are the ones that serve every step): record it from this step's Python walk
                rec = Recorder()
                rec.tail_python = True
                self.backend.tracer = rec
            try:
                env._reset_done(tm._terminated_buf, tm._truncated_buf)
                obs_tail = env.get_observations()
                ro = getattr(env, "_rollout", None)
                if ro is not None:
                    pol = next((m for m in env.managers["observation"] if m.name == ro.obs_name), None)
                    ro.write(pol._last_out if pol is not None else obs_tail, rm._reward_buf if rm is not None else env._reward_buf,
                             tm._terminated_buf, tm._truncated_buf)
            finally:
                env.stats.ptr_override, env._in_step, env._tail_trace = None, False, None
                if rec is not None:
                    self.backend.tracer = None
                    if rec.tail.get("reset"):      # only a step that reset an env can yield both parts: the others do not count
                        self._tail_tries += 1
                        self._adopt_tail(rec.tail)
        env._finish_step_light(snap)
        extras = env._extras
        obs = extras["observations"].get("policy") if len(env.managers["observation"]) > 0 else obs_tail
        if self._env_obs_python:
            obs = env._step_obs   # what the env's own get_observations() returned (a python phase behind the last launch)
        elif self._env_obs_tail:
            obs = obs_tail
        return obs, rm._reward_buf if rm is not None else env._reward_buf, tm._terminated_buf, tm._truncated_buf, extras


def traceable(env, tail_python: bool = False) -> bool:
    """Static conditions under which an env's step may be recorded (see module docstring)."""
    from .managed_env import ManagedEnvironment, _most_derived_is_ours
    from .managers.action import PositionActionManager

    # (a step() override is user code AROUND the step — `…; return super().step(actions)` — i.e. code of the training loop: what is
    # recorded is ManagedEnvironment.step itself, and a manager call the override makes between steps goes through the same
    # watched-descriptor check as one the training script makes, StepTrace.fresh)
    if (type(env).reset is not ManagedEnvironment.reset) != tail_python:
        return False
    if tail_python and env.stats.group is not None:
        return False  # the per-step pack / all-reduce of a process group closes the statistics before the Python tail adds to them
    # (a get_observations() override of the env is user code BEHIND the step's native phases: the recording keeps its place and calls
    #  it again there — ManagedEnvironment._user_get_observations; round 4)
    if not getattr(env.scene, "gf_static_buffers", False) and env._adapter is None:
        return False
    if env._draws:
        return False
    am, tm, rm = env.managers["action"], env.managers["termination"], env.managers["reward"]
    if am is None or tm is None or not isinstance(am, PositionActionManager) or not am.enabled or not tm.enabled:
        return False
    # User-defined manager classes.  The step() / reset(ids) of a user's entity, contact or command manager is code BETWEEN native
    # phases: the recording keeps its place (Recorder.python) and the replay calls it there, exactly as the ordinary step does.
    # Overrides that produce the step's native outputs themselves (action, termination, reward, observation managers) are not
    # something a recording can stand in for.
    between = set(map(id, env.managers["entity"] + env.managers["contact"] + env.managers["command"]))
    # Round 4: the step() of a user TerminationManager / RewardManager class as well — `super().step()` + torch on the manager's buffers
    # is user code at the place of its phase; the launches it makes itself are its own (pointed at the step's statistics slot by the
    # replay) and the native phases behind it read the manager's buffers as they find them.  (A reset() override of these two stays out:
    # their reset is a section of the masked reset.)
    phase_step = set(map(id, [m for m in (tm, rm) if m is not None]))
    # … and the get_observations() / _perform_observation() of a user ObservationManager class: user code at the manager's place
    # among the observations (ManagedEnvironment._observe_all), the manager simply is not part of the recording.
    phase_obs = set(map(id, env.managers["observation"]))
    user_obs = set()
    # … and the step() / handle_actions() of a user action manager class (the reference's extension point,
    # position_action_manager.py:389-392): the env's bookkeeping launch takes the action kernel's place in the recording (it keeps the
    # statistics ring), the user's code — `super().step(...)` launch included — follows it as a python phase (on a Genesis-shaped scene
    # in front of StepTrace._scene_pre, which then leaves sending the targets to it).
    phase_act = {id(am)}
    # On a Genesis-shaped scene a reset(ids) override of a manager that writes SIMULATOR state (action: joint positions, entity: base
    # pose) goes through the envs_idx setters in the middle of the step; the phases behind it would read this tick's snapshot, which
    # those setters do not touch: such a step stays ordinary (every getter call fetches again).
    writes_sim = set(map(id, env.managers["entity"] + [am])) if env._adapter is not None else set()
    for m in env._all_managers() + env.managers["terrain"]:
        for meth in ("step", "reset", "get_observations", "_perform_observation", "handle_actions"):
            if hasattr(m, meth) and not _most_derived_is_ours(m, meth):
                if meth == "reset" and id(m) in writes_sim:
                    return False
                if id(m) in between and meth in ("step", "reset"):
                    continue
                if id(m) in phase_step and meth == "step":
                    continue
                if (id(m) in phase_step or id(m) in phase_act) and meth == "reset":
                    continue   # (a manager with its own reset() is reset by index list behind the masked reset: _reset_partition, _indexed_reset)
                if id(m) in phase_act and meth in ("step", "handle_actions"):
                    continue
                if id(m) in phase_obs and meth in ("get_observations", "_perform_observation"):
                    user_obs.add(id(m))
                    continue
                return False
    if tm._dirty or tm._program.slots.volatile:
        return False
    if rm is not None and (not rm.enabled or rm._dirty or rm._program.slots.volatile):
        return False
    for cm in env.managers["command"]:
        if cm._external_controller is not None or not cm.enabled:
            return False
    for c in env.managers["contact"]:
        if not c.enabled:
            return False
    if tail_python:
        return True  # reset and observations run phase by phase unless StepTrace._tail_native_ok() lets their launches be replayed
    for om in env.managers["observation"]:
        if id(om) not in user_obs and not om._traceable():
            return False
    for em in env.managers["entity"]:
        if not em.enabled or not em._can_fuse_reset():
            return False
    if not am._can_fuse_reset():
        return False
    return True
