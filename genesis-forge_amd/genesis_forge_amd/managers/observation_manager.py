"""
ObservationManager — API mirror of genesis_forge/managers/observation_manager.py.

``get_observations`` (:218-256: per item fn → in-place scale → uniform noise → cat, then the history
list pop/insert/cat) is one ``gf_observe`` launch that writes the final ``[N, O*H]`` tensor.

How opaque getter lambdas get fused: the example configs write observation items as
``lambda env: self.robot_manager.get_angular_velocity()``.  At ``build()`` every item fn is called
once (the reference does the same trial observation to size the space, :182-216).  Tensors returned by
the library's getters carry a provenance tag; if an item's result *is* such a tagged tensor the item
is compiled to the matching opcode and its Python fn is never called again.  Anything else (a lambda
that post-processes, user manager methods) stays an EXTERNAL item evaluated by Python each step.
Pass ``fused=False`` to force the EXTERNAL path for every item.

Returned tensors (``output=``).  The reference returns a fresh ``torch.cat`` every call and keeps its history list private
(:218-226), so a caller may hold ``obs`` for as long as it likes and edit it in place.  ``output="fresh"`` (the default)
keeps exactly that contract: the kernel writes into buffers only this manager sees and the caller gets a copy.
``output="static"`` is the opt-in fast path in the style of CUDA/HIP-graph replays with static outputs: the returned tensor
IS one of ``static_slots`` (3) persistent buffers the kernel writes in rotation — valid until this manager has been asked for
``static_slots - 1`` further observations, read-only for the caller when ``history_len > 1`` (the next call reads its history
frames from it).  No copy, no allocation, nothing that changes from step to step in a recorded step.
With ``history_len > 1`` both modes have two ways to produce the ``[N, H*O]`` tensor (``history=``, default ``"auto"``): ``"shift"`` —
the observation launch copies the first ``H-1`` frames of its previous output behind the new frame (one launch, ``(2H-1)*O`` floats
per env moved inside the step's kernel) — or ``"unroll"``: the manager keeps the history as an in-place ring, the step's kernel
writes only the new frame, and ``gf_history_unroll`` gathers the ring into the newest-first tensor as a streaming launch of its
own (``csrc/gf_unroll.hip``).  In ``"fresh"`` mode the gather writes straight into the caller's new tensor, so it REPLACES the
clone (gait task, 65 536 envs: 194 → 157 µs per step); ``"auto"`` = unroll for ``"fresh"``, shift for ``"static"`` (there the gather
is an extra launch with the same traffic as the in-kernel shift and measures the same or slower: 145 vs 141 µs, 47.6 vs 42.2 µs
at 8 192 envs).
``output="ring"`` (``history_len > 1`` only) is the in-place history ring: ONE persistent ``[N, H, O]`` buffer is the history,
a call writes only the new frame into frame slot ``history_head`` and returns the buffer as ``[N, H*O]`` — ``O`` floats of
traffic per env instead of the ``(2H-1)*O`` a newest-first concatenation costs (gait task: 20 MB instead of 182 MB per step at
65 536 envs).  The frames are the reference's, their ORDER is by slot: newest first is ``history_order()`` (slots from
``history_head`` upwards, wrapping); ``ordered(obs)`` gathers a tensor into the reference's layout for consumers that need it.
``output="window"`` (``history_len > 1`` only; round 4) gives the reference's LAYOUT at the ring's cost: the history lives in ONE
persistent ``[N, S + H - 1, O]`` buffer (``S = H + window_slack`` slots are cycled through, downwards; the last ``H - 1`` slots mirror
the first ``H - 1`` so that a window never wraps), a call writes only the new frame and returns ``H`` consecutive slots of it as an
``[N, H*O]`` strided VIEW — newest frame first, exactly the reference's concatenation, no gather and no copy (``O`` floats per env and
call, plus ``2(H-1)·O / S`` for the mirror copy once per cycle; rows are contiguous, the row stride is ``(S + H - 1)·O``, which
``torch.nn.functional.linear`` and ``copy_`` take as they are).  The contract: a returned tensor stays intact for ``window_slack + 1``
further observations of this manager (default ``H + 1``: enough for an RL loop that stores ``obs_t`` after ``step t + 1``, as rsl_rl's
does), then its oldest frames are overwritten; it is read-only for the caller.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Callable, Optional, TypedDict

import numpy as np
import torch

from .. import _native as nat
from .. import gs
from ..spaces import Box
from ._program import _Slots, _col, call_untraced
from .base import BaseManager, LiveAttr
from .config import ObservationConfigItem


class ObservationConfig(TypedDict):
    fn: Callable[..., torch.Tensor]
    params: dict[str, Any]
    scale: float
    noise: float


_SRC_OPS = {
    "ang_vel": nat.GF_O_ANG_VEL_BODY, "lin_vel": nat.GF_O_LIN_VEL_BODY, "grav": nat.GF_O_PROJ_GRAVITY,
    "dof_pos": nat.GF_O_DOF_POS, "dof_vel": nat.GF_O_DOF_VEL, "dof_force": nat.GF_O_DOF_FORCE,
    "actions": nat.GF_O_ACTIONS, "raw_actions": nat.GF_O_RAW_ACTIONS, "cmd": nat.GF_O_COMMAND,
    "contact_norm": nat.GF_O_CONTACT_FORCE_NORM,
}

_OBS_RING = 3  # persistent output slots: output="static" hands them out directly (valid for two further calls)


class ObservationManager(BaseManager):
    noise = LiveAttr("noise")   # manager-wide noise, read per step in the reference (observation_manager.py:246-250)
    #: what get_observations() returns when the constructor is not told: "fresh" (reference contract) or "static" (see module
    #: docstring).  GF_OBS_OUTPUT overrides the default for a whole process (unchanged task configs on the fast path).
    default_output = os.environ.get("GF_OBS_OUTPUT", "fresh")
    #: how a history of H > 1 frames becomes the [N, H*O] tensor in the "fresh" / "static" modes: "shift", "unroll" or "auto"
    default_history = os.environ.get("GF_OBS_HISTORY", "auto")
    static_slots = _OBS_RING
    #: output="window": further observations a returned tensor survives, minus one (None: history_len)
    window_slack: Optional[int] = None

    """Generates an observation tensor from a dict of items (ctor as observation_manager.py:134-156)."""

    def __init__(self, env, cfg: dict[str, ObservationConfig], name: str = "policy", history_len: int | None = None,
                 noise: float | None = None, fused: bool = True, output: str | None = None, history: str | None = None):
        super().__init__(env, "observation")
        self._name = name
        self._output = output if output is not None else type(self).default_output
        if self._output not in ("fresh", "static", "ring", "window"):
            raise ValueError("output must be 'fresh', 'static', 'ring' or 'window'")
        self._history_mode = history if history is not None else type(self).default_history
        if self._history_mode not in ("auto", "shift", "unroll"):
            raise ValueError("history must be 'auto', 'shift' or 'unroll'")
        self._unroll_args = nat.GfHistoryUnrollArgs()
        self._unroll_out: Optional[torch.Tensor] = None
        self._ring: Optional[torch.Tensor] = None   # the [N, H, O] history of the "unroll" strategy
        self._fresh_pool: list = []                 # output="fresh": (tensor, address) rows of the current block not handed out yet
        self._fresh_ptr = 0
        self._unrolled = False       # history kept as a ring + gather launch (module docstring: ``history=``); set by _refresh_modes()
        self._direct_fresh = False   # output="fresh" without history: the launch writes straight into the caller's new tensor
        self._frame_only = False     # inside _perform_observation(): the launch writes the frame alone
        self._history: Optional[list] = None   # the reference's frame list — only for a subclass that overrides _perform_observation()
        self._window = False         # output="window" with a history: the returned tensor is a strided view of _win
        self._win: Optional[torch.Tensor] = None    # [N, (S + H - 1) * O]
        self._win_view: Optional[torch.Tensor] = None
        self._win_cycle = self._win_slots = 0       # S (slots cycled through), S + H - 1 (slots per env incl. the mirror tail)

        self.noise = noise
        self._observation_size = 1
        self._observation_space = None
        self._fused = fused
        if history_len is not None and history_len < 1:
            raise ValueError("history_len must be greater than 0")
        self._history_len = history_len if history_len is not None else 1
        if len(cfg) > nat.GF_MAX_OBS_ITEMS:
            raise ValueError(f"ObservationManager supports at most {nat.GF_MAX_OBS_ITEMS} items")
        self.cfg: dict[str, ObservationConfigItem] = {}
        for item_name, c in cfg.items():
            self.cfg[item_name] = ObservationConfigItem(c, env, on_dirty=self._mark_dirty)
        self._dirty = True
        self._plan: list = []
        self._args = nat.GfObservationArgs()
        self._bufs: list[torch.Tensor] = []
        self._rotor = nat.GfRotor()           # which output slot is current (shared with a recorded step's native patch table)
        self._ring_clock = nat.GfRingClock()  # observations produced in ring mode (slot = (H - calls % H) % H)

    def _mark_dirty(self, soft: bool = False):
        """``soft`` (an item's scale / noise / a param value was assigned): the item table is compiled again into the SAME descriptor
        before the next step (``_compile`` writes ``self._args`` in place), a recorded step goes on (RewardManager._mark_dirty)."""
        self._dirty = True
        env = self.env
        if soft and self._bufs and hasattr(env, "_soft_dirty"):
            env._soft_dirty.add(self)
        elif hasattr(env, "invalidate_trace"):
            env.invalidate_trace()

    def _live_attr_changed(self, name: str = "") -> None:
        if name == "noise":   # the manager-wide noise level is a number of the item table, like an item's own (a noise curriculum)
            self._mark_dirty(soft=True)
        else:
            super()._live_attr_changed(name)

    def _refresh_in_place(self) -> bool:
        if not self.enabled or not self._bufs:
            return False
        before = [(self._args.items[k].op, self._args.items[k].i0, self._args.items[k].i1, self._args.items[k].width) for k in range(self._args.num_items)]
        exts = len(self._slots.exts)
        self._compile()
        a = self._args
        now = [(a.items[k].op, a.items[k].i0, a.items[k].i1, a.items[k].width) for k in range(a.num_items)]
        return now == before and len(self._slots.exts) == exts   # (anything else: the caller drops the recording)

    @property
    def name(self) -> str:
        return self._name

    @property
    def output(self) -> str:
        """"fresh", "static", "ring" or "window" (module docstring).  Assigning it drops a recorded step: the step's patch table is per mode."""
        return self._output

    @output.setter
    def output(self, value: str) -> None:
        if value not in ("fresh", "static", "ring", "window"):
            raise ValueError("output must be 'fresh', 'static', 'ring' or 'window'")
        if value != self._output:
            self._output = value
            self._refresh_modes()
            if hasattr(self.env, "invalidate_trace"):
                self.env.invalidate_trace()

    @property
    def observation_space(self):
        return self._observation_space

    # -- build ----------------------------------------------------------------------------------------
    def build(self):
        """Trial observation → item plan, observation space, history buffers (observation_manager.py:182-216)."""
        if not self.enabled:
            self._observation_size = 1
            self._observation_space = Box(low=-np.inf, high=np.inf, shape=(1,), dtype=np.float32)
            return
        env = self.env
        self._plan = []
        width = 0
        for name, cfg in self.cfg.items():
            cfg.build()
            assert callable(cfg.fn), f"Observation function {name} is not callable"
            value = self._call_item(name, cfg)
            src = getattr(value, "_gf_src", None) if self._fused else None
            w = int(value.shape[-1]) if value.dim() > 1 else 1
            self._plan.append((name, cfg, src, w))
            width += w
        if width >= nat.GF_MAX_OBS_WIDTH:
            raise ValueError(f"observation frame wider than {nat.GF_MAX_OBS_WIDTH - 1}")
        self._frame = width
        self._observation_size = width * self._history_len
        self._observation_space = Box(low=-np.inf, high=np.inf, shape=(self._observation_size,), dtype=np.float32)
        self._bufs = [torch.zeros((env.num_envs, self._observation_size), device=gs.device, dtype=gs.tc_float) for _ in range(_OBS_RING)]
        self._rotor.cur, self._rotor.count = 0, _OBS_RING
        for i, b in enumerate(self._bufs):
            self._rotor.slot[i] = b.data_ptr()
        self._ring_clock.calls, self._ring_clock.length = 0, self._history_len
        self._ring = None
        self._win = self._win_view = None
        self._history = None
        self._fresh_pool = []
        self._dirty = True
        self._refresh_modes()

    def _refresh_modes(self) -> None:
        """The two per-mode flags the step's hot path reads (plain attributes: a property costs the host ~0.25 µs per read and a
        recorded step reads them seven times)."""
        built = bool(self._bufs)
        H = self._history_len
        # (a window over a history of one frame is that frame: the plain fresh tensor)
        self._direct_fresh = built and self._output in ("fresh", "window") and H == 1
        self._window = built and self._output == "window" and H > 1
        if self._window and self._win is None:
            slack = H if self.window_slack is None else max(1, int(self.window_slack))
            self._win_cycle, self._win_slots = H + slack, 2 * H + slack - 1
            self._win = torch.zeros((self.env.num_envs, self._win_slots * self._frame), device=gs.device, dtype=gs.tc_float)   # zero history, as the reference's
            self._ring_clock.calls = 0
        if built:
            self._ring_clock.length = self._win_cycle if self._window else H
        on = built and H > 1 and self._output not in ("ring", "window") and (
            self._history_mode == "unroll" or (self._history_mode == "auto" and self._output == "fresh"))
        self._unrolled = on
        if on and self._ring is None:
            self._ring = torch.zeros_like(self._bufs[0])   # [N, H, O], zero history like the reference's initial frame list
            u = self._unroll_args
            u.ring, u.out2, u.num_envs = self._ring.data_ptr(), None, self.env.num_envs
            u.frame_width, u.history_len, u.ring_slot = self._frame, self._history_len, 1
            self._unroll_out = self._bufs[0]

    def _call_item(self, name, cfg) -> torch.Tensor:
        try:
            return cfg.fn(env=self.env, **cfg.params)
        except Exception as e:  # observation_manager.py:253-255
            print(f"Error generating observation for '{name}'")
            raise e

    def _compile(self):
        a = self._args
        self._slots = _Slots(self.env)
        self._entity = None
        self._entity_mgr = None
        self._am = None
        n = 0
        for name, cfg, src, w in self._plan:
            it = a.items[n]
            it.width = w
            scale = cfg.scale
            it.scale = 1.0 if scale is None else float(scale)
            noise = cfg.noise or self.noise
            it.noise = 0.0 if noise is None else float(noise)
            kind = src[0] if src is not None else None
            owner = src[1] if src is not None else None
            fused = kind in _SRC_OPS
            if fused and kind in ("ang_vel", "lin_vel", "grav"):
                ent = owner.entity
                if self._entity is None:
                    self._entity = ent
                    self._entity_mgr = owner
                fused = ent is self._entity
            if fused and kind in ("dof_pos", "dof_vel", "dof_force", "actions"):
                if self._am is None:
                    self._am = owner
                fused = owner is self._am
            if fused:
                it.op = _SRC_OPS[kind]
                if kind == "cmd":
                    it.i0 = self._slots.cmd(owner)
                elif kind == "contact_norm":
                    it.i0 = self._slots.contact(owner, False)
            else:
                it.op = nat.GF_O_EXTERNAL
                it.i0 = self._slots.ext(lambda name=name, cfg=cfg: self._call_item(name, cfg))
                it.i1 = w
            n += 1
        a.num_items = n
        a.obs_width = self._frame
        a.history_len = self._history_len
        a.num_envs = self.env.num_envs
        self._dirty = False

    # -- public ---------------------------------------------------------------------------------------
    def get_observations(self) -> torch.Tensor:
        """observation_manager.py:218-226 → one launch.  A subclass that overrides ``_perform_observation()`` (the reference's
        get_observations() calls it for the new frame) gets the reference's own sequence instead: frame list, ``torch.cat``."""
        env = self.env
        if not self.enabled:
            return torch.zeros((env.num_envs, self._observation_size))
        if type(self)._perform_observation is not ObservationManager._perform_observation:
            if self._history is None:
                shape = (env.num_envs, self._frame)
                self._history = [torch.zeros(shape, device=gs.device, dtype=gs.tc_float) for _ in range(self._history_len)]
            self._history.pop()
            self._history.insert(0, self._perform_observation())
            return torch.cat(self._history, dim=-1)
        return self._observe()

    def _perform_observation(self) -> torch.Tensor:
        """observation_manager.py:232-256: one round of observations — the ``[N, O]`` frame, no history (one launch into a new tensor)."""
        if self._dirty:
            self._compile()
        self._frame_only = True
        try:
            return self._observe()
        finally:
            self._frame_only = False
            self._args.history_len = self._history_len

    def _observe(self) -> torch.Tensor:
        env = self.env
        if self._dirty:
            self._compile()
        a = self._args
        keep: list = []
        if self._entity is not None:
            env.entity_views(self._entity).fill(a.entity)
            st = self._entity_mgr.stale()
            if st is not None:
                keep.extend(st)
                a.stale_quat = st[0].data_ptr()
                a.stale_mask = st[1].data_ptr()
                a.stale_mask2 = None if st[2] is None else st[2].data_ptr()
            else:
                a.stale_quat = a.stale_mask = a.stale_mask2 = None
        if self._am is not None:
            am = self._am
            a.num_dofs = am.num_actions
            ops = {a.items[k].op for k in range(a.num_items)}
            if nat.GF_O_DOF_POS in ops:
                t = am._scene_dofs("position"); keep.append(t); a.dof_pos = t.data_ptr()
            if nat.GF_O_DOF_VEL in ops:
                t = am._scene_dofs("velocity"); keep.append(t); a.dof_vel = t.data_ptr()
            if nat.GF_O_DOF_FORCE in ops:
                t = am._scene_dofs("force"); keep.append(t); a.dof_force = t.data_ptr()
            a.targets = am._actions.data_ptr()
        if env.actions is not None:
            a.env_actions = env.actions.data_ptr()
        # ext columns are [N, w]
        n = env.num_envs
        for k, src in enumerate(self._slots.cmds):
            if hasattr(src, "_gf_command_view"):
                src._gf_command_view(a.command[k])
                continue
            t = src.command
            t = _col(t.unsqueeze(-1) if t.dim() == 1 else t, n, torch.float32)
            keep.append(t)
            a.command[k].command, a.command[k].width, a.command[k].stride = t.data_ptr(), t.shape[1], 0
        for k, (mgr, lv, _lp) in enumerate(self._slots.contacts):
            keep.extend(mgr.view(a.contact[k], need_link_vel=False))
        self._bind_exts(a, keep)
        draws = env.take_draws(f"obs:{self._name}")
        keep.append(draws)
        a.noise_draws = None if draws is None else draws.data_ptr()
        a.seed, a.stream, a.env_offset = env._rng_seed, env.next_stream(), env.env_offset
        out = self._rotate_ring(a)
        env.backend.call("observe", a, owner=self)
        self._keep = keep
        if self._frame_only:
            return out
        if self._unrolled:   # the launch wrote the new frame into the ring: gather the ring into this call's tensor
            u = self._unroll_args
            u.ring_slot, u.out2 = a.history_ring, None
            out = self._next_unroll_out()
            env.backend.call("history_unroll", u, owner=self)
        self._last_out = out    # the buffer the kernel wrote (what a rollout storage copies from)
        return self._hand_out(out)

    def _hand_out(self, out: torch.Tensor) -> torch.Tensor:
        """The caller's tensor: one nobody else holds (reference contract) — the launch wrote it directly (no history: a new tensor
        per call; history kept as a ring: the gather's destination), or it is a copy — or the persistent slot itself (static)."""
        return out.clone() if self._output == "fresh" and not self._unrolled and not self._direct_fresh else out

    # -- history window (output="window") ----------------------------------------------------------------------------
    def _window_at(self, slot: int) -> torch.Tensor:
        """H consecutive frame slots from ``slot`` on, as the reference's [N, H*O] newest-first tensor (a strided view)."""
        O = self._frame
        return torch.as_strided(self._win, (self._win.shape[0], self._history_len * O), (self._win_slots * O, 1), slot * O)

    def _mirror_tail(self) -> None:
        """The slot walk wraps from 0 to S - 1: the newest H - 1 frames (slots 0 … H-2) are copied behind slot S - 1, so that the
        windows of the next H - 1 calls are consecutive slots too.  One strided copy per cycle."""
        H, S = self._history_len, self._win_cycle
        w = self._win.view(self._win.shape[0], self._win_slots, self._frame)
        w[:, S:S + H - 1].copy_(w[:, :H - 1])

    def _window_slot(self, calls: int) -> int:
        S = self._win_cycle
        return (S - calls % S) % S

    def _take_fresh(self) -> torch.Tensor:
        """A tensor nobody else holds.  Observations come out of blocks of 3 … 32 rows (≈ 16 MB; from 64 MB per observation on:
        one allocation per call): ONE ``torch.empty`` and ONE ``unbind`` per block, the rows' addresses computed — per step the host
        pops a (tensor, address) pair.  An allocator call per step costs ~2.5 µs, a slice + view per row ~3 µs: both show at 4 096
        envs, where the host has ~13 µs per step (20.9 / 19.9 instead of 16.3 µs).  Each row is a separate tensor, never handed out
        twice; a block's memory goes back to the allocator when the last of its rows has been dropped.  Rows start 256-byte aligned
        (the kernels store 16-byte units): the env count is padded to a multiple of 64 inside the block."""
        pool = self._fresh_pool
        if not pool:
            ref = self._bufs[0]
            n, w = ref.shape
            n_pad = (n + 63) & ~63
            nbytes = n_pad * w * ref.element_size()
            k = 1 if nbytes >= (64 << 20) else max(3, min(32, (16 << 20) // nbytes))
            block = torch.empty((k, n_pad, w), dtype=ref.dtype, device=ref.device)
            rows = block.unbind(0)
            if n_pad != n:
                rows = [r[:n] for r in rows]
            base = block.data_ptr()
            pool = self._fresh_pool = [(r, base + i * nbytes) for i, r in enumerate(rows)]
        out, self._fresh_ptr = pool.pop()
        return out

    def _next_fresh_out(self, a) -> torch.Tensor:
        out = self._take_fresh()
        a.obs = self._fresh_ptr
        self._unroll_out = out
        return out

    def _next_unroll_out(self) -> torch.Tensor:
        """Destination of this call's gather: a new tensor the caller will own ("fresh"), or the next static slot."""
        if self._output == "fresh":
            out = self._take_fresh()
            self._unroll_args.out = self._fresh_ptr
        else:
            ro = self._rotor
            ro.cur = (ro.cur + 1) % _OBS_RING
            out = self._bufs[ro.cur]
            self._unroll_args.out = out.data_ptr()
        self._unroll_out = out
        return out

    # -- in-place history ring (output="ring") ------------------------------------------------------------------
    @property
    def _in_place(self) -> bool:
        return self._output == "ring" and self._history_len > 1

    @property
    def history_head(self) -> int:
        """Frame slot of the newest frame in the ``output="ring"`` buffer (after the most recent observation)."""
        H = self._history_len
        calls = self._ring_clock.calls
        return (H - (calls - 1) % H) % H if calls else 0

    def history_order(self) -> list:
        """Frame slots newest first: ``history_head``, then upwards, wrapping."""
        H, h = self._history_len, self.history_head
        return [(h + k) % H for k in range(H)]

    def ordered(self, obs: torch.Tensor) -> torch.Tensor:
        """``obs`` (an ``output="ring"`` tensor) gathered into the reference's newest-first layout (a copy)."""
        if not self._in_place:
            return obs
        n, H, O = obs.shape[0], self._history_len, self._frame
        return obs.view(n, H, O)[:, self.history_order(), :].reshape(n, H * O)

    def _rotate_ring(self, a) -> torch.Tensor:
        if self._frame_only:   # _perform_observation(): the frame alone, into a tensor of the caller's
            a.history_len, a.history_ring, a.ring_slots, a.prev_obs = 1, 0, 0, None
            out = torch.empty((self.env.num_envs, self._frame), device=gs.device, dtype=gs.tc_float)
            a.obs = out.data_ptr()
            return out
        if self._window:
            ck = self._ring_clock
            slot = self._window_slot(ck.calls)
            if ck.calls and slot == self._win_cycle - 1:
                self._mirror_tail()
            a.history_ring, a.ring_slots = slot + 1, self._win_slots
            ck.calls += 1
            a.prev_obs, a.obs = None, self._win.data_ptr()
            self._win_view = self._window_at(slot)
            return self._win_view
        a.ring_slots = 0
        if self._in_place or self._unrolled:
            H, ck = self._history_len, self._ring_clock
            a.history_ring = (H - ck.calls % H) % H + 1
            ck.calls += 1
            a.prev_obs = None
            ring = self._ring if self._unrolled else self._bufs[0]
            a.obs = ring.data_ptr()
            return ring
        a.history_ring = 0
        if self._direct_fresh:
            a.prev_obs = None
            return self._next_fresh_out(a)
        ro = self._rotor
        prev = self._bufs[ro.cur]
        ro.cur = (ro.cur + 1) % _OBS_RING
        out = self._bufs[ro.cur]
        a.prev_obs = prev.data_ptr() if self._history_len > 1 else None
        a.obs = out.data_ptr()
        return out

    def _current_out(self) -> torch.Tensor:
        if self._window:
            return self._win_view
        if self._unrolled or self._direct_fresh:
            return self._unroll_out
        return self._bufs[0] if self._in_place else self._bufs[self._rotor.cur]

    def _bind_exts(self, a, keep: list) -> None:
        """Evaluate the items no kernel opcode covers (user callables) and hand their [N,w] columns to the descriptor."""
        n = self.env.num_envs
        for k, prov in enumerate(self._slots.exts):
            t = call_untraced(self.env, prov)
            t = _col(t.unsqueeze(-1) if t.dim() == 1 else t, n, torch.float32)
            keep.append(t)
            a.ext[k] = t.data_ptr()

    def _traceable(self) -> bool:
        return self.enabled and not self._dirty and all(
            not hasattr(s, "_external_controller") or s._external_controller is None for s in self._slots.cmds)

    def _trace_pre(self, args):
        """Recorded step: Python-level items are evaluated right before this manager's op, where the ordinary path calls them."""
        if not self._slots.exts or args is self._unroll_args:   # (the gather op of a ring-kept history has no items)
            return None

        def pre(self=self, a=args):
            keep: list = []
            self._bind_exts(a, keep)
            self._keep_ext = keep

        return pre

    def _trace_native(self, args) -> list:
        """Recorded step: the per-step fields of this manager's launch as native patches (GfReplayPatch) — the noise stream and
        the output slot rotation / ring slot, exactly what the ordinary path does in get_observations()."""
        P = nat.GfReplayPatch
        out = [P(nat.GF_PATCH_STREAM, 0, nat.field_addr(args, "stream"), None, None)]
        if self._unrolled:   # the frame goes into the ring; the gather op that follows learns the slot from the same patch
            args.prev_obs, args.obs = None, self._ring.data_ptr()
            out.append(P(nat.GF_PATCH_RING_SLOT, 0, nat.field_addr(args, "history_ring"), nat.field_addr(self._unroll_args, "ring_slot"),
                         C.addressof(self._ring_clock)))
        elif self._window:   # the slot walks down a cycle of S slots (the clock's length); the view follows in _trace_after
            args.prev_obs, args.obs, args.ring_slots = None, self._win.data_ptr(), self._win_slots
            out.append(P(nat.GF_PATCH_RING_SLOT, 0, nat.field_addr(args, "history_ring"), None, C.addressof(self._ring_clock)))
        elif self._in_place:
            args.prev_obs, args.obs = None, self._bufs[0].data_ptr()
            out.append(P(nat.GF_PATCH_RING_SLOT, 0, nat.field_addr(args, "history_ring"), None, C.addressof(self._ring_clock)))
        elif self._direct_fresh:
            args.history_ring, args.prev_obs = 0, None   # `obs` = this step's new tensor (_trace_fresh_patch)
        else:
            args.history_ring = 0
            out.append(P(nat.GF_PATCH_ROTATE, 0, nat.field_addr(args, "prev_obs") if self._history_len > 1 else None,
                         nat.field_addr(args, "obs"), C.addressof(self._rotor)))
        return out

    def _trace_fresh_patch(self, args):
        """Recorded step, output="fresh" without history: a Python patch that points the launch at this step's new tensor."""
        if self._window:
            # … output="window": the mirror copy of the step whose slot wraps (enqueued in front of the step's launches)
            def mirror(_actions, self=self, ck=self._ring_clock):
                if ck.calls and ck.calls % ck.length == 1:
                    self._mirror_tail()

            return mirror
        if not self._direct_fresh:
            return None

        def patch(_actions, self=self, a=args):
            self._next_fresh_out(a)

        return patch

    def _trace_unroll(self, args):
        """Recorded step, the gather op: (Python patch or None, native patches) that give it this step's destination."""
        assert args is self._unroll_args
        if self._output == "fresh":
            def patch(_actions, self=self):
                self._next_unroll_out()

            return patch, []
        return None, [nat.GfReplayPatch(nat.GF_PATCH_ROTATE, 0, None, nat.field_addr(args, "out"), C.addressof(self._rotor))]

    def _trace_after(self) -> None:
        """Recorded step, after the launches have been enqueued: publish this step's observation (a fresh copy by default)."""
        if self._window:   # (the native patch has advanced the clock: the frame of this step sits in the slot of call `calls - 1`)
            self._win_view = self._window_at(self._window_slot(self._ring_clock.calls - 1))
        out = self._bufs[self._rotor.cur] if self._unrolled and self._output == "static" else self._current_out()
        self._unroll_out = out if self._unrolled or self._direct_fresh else self._unroll_out
        self._last_out = out
        self.env._extras["observations"][self._name] = self._hand_out(out)
