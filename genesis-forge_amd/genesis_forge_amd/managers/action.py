"""
Action managers — API mirror of genesis_forge/managers/action/{base.py, position_action_manager.py,
position_within_limits.py}; the per-step math runs in ``gf_action_step`` (Phase A, one launch).

What stays Python (config time only): regex joint selection → ``dofs_idx`` (position_action_manager.py:
300-309), the ``DofValue`` pattern tables → per-DOF tensors (:311-374, :470-513), PD gain upload on reset.
What became one kernel: NaN/Inf scan, ``a*scale+offset``, ``clamp``, and — when driven by
``ManagedEnvironment.step`` — the env bookkeeping of genesis_env.py:196-203 as well.
"""
from __future__ import annotations

import re
from typing import Any, Callable, Optional, TypeVar

import numpy as np
import torch

from .. import _native as nat
from .. import gs
from ..spaces import Box
from .base import BaseManager

T = TypeVar("T")
DofValue = Any  # scalar or {joint-name regex: value}


def _tag(t: torch.Tensor, src) -> torch.Tensor:
    """Provenance tag read by ObservationManager.build() to fuse getter lambdas (see observation_manager.py)."""
    t._gf_src = src
    return t


def _ensure_dof_pattern(value):
    """Scalar → ``{".*": value}`` (position_action_manager.py:18-40)."""
    if value is None:
        return None
    if isinstance(value, dict):
        return value
    return {".*": value}


class BaseActionManager(BaseManager):
    """Base for managers that handle actions (action/base.py:9-102).  ``delay_step`` is accepted and, as in
    the reference (quirk q3: the FIFO is never primed), has no effect."""

    def __init__(self, env, delay_step: int = 0):
        super().__init__(env, type="action")
        self._raw_actions = None
        self._actions = None
        self._delay_step = delay_step
        self._action_delay_buffer: list = []

    @property
    def num_actions(self) -> int:
        return 0

    @property
    def action_space(self):
        return Box(low=-np.inf, high=np.inf, shape=(self.num_actions,), dtype=np.float32)

    @property
    def actions(self) -> torch.Tensor:
        if self._actions is None:
            return torch.zeros((self.env.num_envs, self.num_actions))
        return self._actions

    @property
    def raw_actions(self) -> torch.Tensor:
        if self._raw_actions is None:
            return torch.zeros((self.env.num_envs, self.num_actions))
        return self._raw_actions

    def step(self, actions: torch.Tensor) -> torch.Tensor:
        self._raw_actions = actions
        return actions

    def reset(self, envs_idx):
        pass

    def get_actions(self) -> torch.Tensor:
        if self._actions is None:
            return torch.zeros((self.env.num_envs, self.num_actions))
        return self._actions


class PositionActionManager(BaseActionManager):
    """Converts actions to DOF position targets (ctor args as position_action_manager.py:148-167)."""

    _mode = nat.GF_ACTION_POSITION
    _fused_reset = True

    def __init__(
        self,
        env,
        joint_names: list[str] | str = ".*",
        default_pos: DofValue = {".*": 0.0},
        scale: DofValue = 1.0,
        clip: DofValue = None,
        offset: DofValue = 0.0,
        use_default_offset: bool = True,
        pd_kp: DofValue = None,
        pd_kv: DofValue = None,
        max_force: DofValue = None,
        damping: DofValue = None,
        stiffness: DofValue = None,
        frictionloss: DofValue = None,
        noise_scale: float = 0.0,
        action_handler: Callable[[torch.Tensor], None] = None,
        quiet_action_errors: bool = False,
        delay_step: int = 0,
    ):
        super().__init__(env, delay_step)
        self._default_pos_cfg = _ensure_dof_pattern(default_pos)
        self._offset_cfg = _ensure_dof_pattern(offset)
        self._scale_cfg = _ensure_dof_pattern(scale)
        self._clip_cfg = _ensure_dof_pattern(clip)
        self._pd_kp_cfg = _ensure_dof_pattern(pd_kp)
        self._pd_kv_cfg = _ensure_dof_pattern(pd_kv)
        self._max_force_cfg = _ensure_dof_pattern(max_force)
        self._damping_cfg = _ensure_dof_pattern(damping)
        self._stiffness_cfg = _ensure_dof_pattern(stiffness)
        self._frictionloss_cfg = _ensure_dof_pattern(frictionloss)
        self._quiet_action_errors = quiet_action_errors
        self._enabled_dof: Optional[dict] = None
        self._noise_scale = noise_scale
        self._use_default_offset = use_default_offset
        self._default_dofs_pos: Optional[torch.Tensor] = None
        self._warned = 0

        if use_default_offset and offset != 0.0:
            raise ValueError("Cannot set both use_default_offset and offset")
        if isinstance(joint_names, str):
            self._joint_name_cfg = [joint_names]
        elif isinstance(joint_names, list):
            self._joint_name_cfg = joint_names
        else:
            raise TypeError(f"Invalid joint_names type: {type(joint_names)}")

    # -- properties ---------------------------------------------------------------------------------
    @property
    def action_space(self):
        return Box(low=-np.inf, high=np.inf, shape=(self.num_actions,), dtype=np.float32)

    @property
    def num_actions(self) -> int:
        assert self._enabled_dof is not None, (
            "PositionalActionManager not initialized. You may need to add <PositionalActionManager>.reset() in your environment's reset method.")
        return len(self._enabled_dof)

    @property
    def dofs_idx(self) -> list[int]:
        return list(self._enabled_dof.values())

    @property
    def default_dofs_pos(self) -> torch.Tensor:
        return self._default_dofs_pos

    # -- DOF getters (position_action_manager.py:243-289) ---------------------------------------------
    def _scene_dofs(self, what: str) -> torch.Tensor:
        """[N,D] f32 contiguous state of the controlled DOFs; zero-copy on the synthetic scene."""
        robot = self.env.robot
        if hasattr(robot, "gf_dofs"):
            return robot.gf_dofs(what, self.dofs_idx)
        ad = self.env._adapter
        if ad is not None:
            return ad.dofs(robot, what, self.dofs_idx)   # Genesis-shaped scene: one getter call per tick, shared by every phase
        t = getattr(robot, "get_dofs_" + what)(self.dofs_idx)
        return t.to(torch.float32).contiguous()

    def get_dofs_position(self, noise: float = 0.0):
        pos = self.env.robot.get_dofs_position(self.dofs_idx)
        if noise > 0.0:
            return self._add_random_noise(pos, noise)
        return _tag(pos, ("dof_pos", self))

    def get_dofs_velocity(self, noise: float = 0.0, clip: tuple[float, float] = None):
        vel = self.env.robot.get_dofs_velocity(self.dofs_idx)
        if noise > 0.0:
            vel = self._add_random_noise(vel, noise)
        if clip is not None:
            vel = vel.clamp(**clip)
        if noise > 0.0 or clip is not None:
            return vel
        return _tag(vel, ("dof_vel", self))

    def get_dofs_force(self, noise: float = 0.0, clip_to_max_force: bool = False):
        force = self.env.robot.get_dofs_force(self.dofs_idx)
        if noise > 0.0:
            force = self._add_random_noise(force, noise)
        clipped = clip_to_max_force and self._force_range is not None
        if clipped:
            force = force.clamp(self._force_range[0], self._force_range[1])
        if noise > 0.0 or clipped:
            return force
        return _tag(force, ("dof_force", self))

    def get_actions(self) -> torch.Tensor:
        """The processed actions == clamped PD targets (quirk q1, position_action_manager.py:385-387)."""
        if self._actions is None:
            return torch.zeros((self.env.num_envs, self.num_actions))
        return _tag(self._actions, ("actions", self))

    # -- build --------------------------------------------------------------------------------------
    #: (attribute, constructor cfg) of the optional per-DOF gain vectors — resolved in one loop instead of a block per gain
    _GAIN_TABLE = (("_scale_values", "_scale_cfg"), ("_kp_values", "_pd_kp_cfg"), ("_kv_values", "_pd_kv_cfg"),
                   ("_damping_values", "_damping_cfg"), ("_stiffness_values", "_stiffness_cfg"), ("_frictionloss_values", "_frictionloss_cfg"))

    def build(self):
        """Resolve the controlled joints and every per-DOF constant (what position_action_manager.py:295-374 computes): revolute joints
        whose name fully matches one of ``joint_names``, in the robot's joint order; then each ``{pattern: value}`` table onto them."""
        def wanted(name):
            return any(re.match(f"^{pattern}$", name) for pattern in self._joint_name_cfg)

        self._enabled_dof = {j.name: j.dof_start for j in self.env.robot.joints if j.type == gs.JOINT_TYPE.REVOLUTE and wanted(j.name)}
        D, N = self.num_actions, self.env.num_envs
        vec = lambda table, **kw: torch.tensor(self._per_dof(table, **kw), device=gs.device, dtype=gs.tc_float)

        self._default_vec = vec(self._default_pos_cfg) if self._default_pos_cfg is not None else torch.zeros(D, device=gs.device)
        self._default_dofs_pos = self._default_vec.unsqueeze(0).expand(N, -1)
        for attr, cfg_attr in self._GAIN_TABLE:
            table = getattr(self, cfg_attr)
            setattr(self, attr, None if table is None else vec(table))

        lower, upper = self.env.robot.get_dofs_limit(self.dofs_idx)
        limits = torch.stack([lower.to(gs.tc_float), upper.to(gs.tc_float)], dim=1)            # [D, 2]: the joint limits …
        self._clip_values = limits if self._clip_cfg is None else vec(self._clip_cfg, start=[list(map(float, r)) for r in limits.tolist()])   # … unless `clip` narrows them

        if self._use_default_offset:
            self._offset_values, self._offset_vec = self._default_dofs_pos, self._default_vec
        else:
            self._offset_vec = vec(self._offset_cfg if self._offset_cfg is not None else {".*": 0.0})
            self._offset_values = self._offset_vec

        # max_force: a bound b means (-b, b), a pair means (lo, hi) — decided, as upstream, by the FIRST joint's entry
        self._force_range = None
        if self._max_force_cfg is not None:
            entries = self._per_dof(self._max_force_cfg)
            paired = isinstance(entries[0], (list, tuple))
            lo = [e[0] if paired else -e for e in entries]
            hi = [e[1] if paired else e for e in entries]
            self._force_range = (torch.tensor(lo, device=gs.device), torch.tensor(hi, device=gs.device))

        self._build_native()

    def _build_native(self):
        """Per-DOF constant vectors of Phase A and the persistent target buffer."""
        D, N = self.num_actions, self.env.num_envs
        scale = self._scale_values if self._scale_values is not None else torch.ones(D, device=gs.device)
        self._k_scale = scale.to(gs.tc_float).contiguous()
        self._k_offset = self._offset_vec.to(gs.tc_float).contiguous()
        self._k_lo = self._clip_values[:, 0].contiguous()
        self._k_hi = self._clip_values[:, 1].contiguous()
        self._k_default = self._default_vec.to(gs.tc_float).contiguous()
        self._actions = torch.zeros((N, D), device=gs.device, dtype=gs.tc_float)
        self._args = nat.GfActionArgs()

    # -- step ---------------------------------------------------------------------------------------
    def step(self, actions: torch.Tensor, _fuse_env: bool = False) -> torch.Tensor:
        """Phase A (position_action_manager.py:376-419).  With ``_fuse_env`` the same launch also performs
        GenesisEnv.step's bookkeeping (genesis_env.py:196-203)."""
        if not self.enabled:
            return
        if self._user_handler():
            # a subclass overrides handle_actions() — the reference's extension point ("Override this function if you want to change
            # the action handling logic", position_action_manager.py:385-392): step() = keep the raw actions, hand them to it
            if actions.dtype != torch.float32 or not actions.is_contiguous():
                actions = actions.to(torch.float32).contiguous()
            self._raw_actions = actions
            out = self.handle_actions(actions)
            if out is not None and out is not self._actions:
                self._actions.copy_(out)   # (the manager's target buffer is persistent: descriptors and the scene address it)
            return self._actions
        return self._process(actions, _fuse_env)

    def _user_handler(self) -> bool:
        return type(self).handle_actions is not PositionActionManager.handle_actions

    def _process(self, actions: torch.Tensor, _fuse_env: bool = False) -> torch.Tensor:
        """The native Phase A launch: scale / offset / clip (or the joint-limit map), NaN / Inf flags, targets to the actuators."""
        env = self.env
        if actions.dtype != torch.float32 or not actions.is_contiguous():
            actions = actions.to(torch.float32).contiguous()
        self._raw_actions = actions
        a = self._args
        a.num_envs, a.num_dofs, a.mode = env.num_envs, self.num_actions, self._mode
        a.check_finite = 0 if self._quiet_action_errors else 1
        a.actions_in = actions.data_ptr()
        a.scale, a.offset = self._k_scale.data_ptr(), self._k_offset.data_ptr()
        a.clip_lo, a.clip_hi = self._k_lo.data_ptr(), self._k_hi.data_ptr()
        if _fuse_env:
            env._ensure_action_buffers(actions)
            a.env_actions, a.env_last_actions = env._actions.data_ptr(), env._last_actions.data_ptr()
            a.episode_length = env.episode_length.data_ptr()
        else:
            a.env_actions = a.env_last_actions = a.episode_length = None
        a.targets = self._actions.data_ptr()
        a.stats = env.stats.ptr if not self._quiet_action_errors else None
        a.stats_zero = a.stats_fold_src = a.stats_fold_dst = a.stats_last_reset = None  # only a recorded step uses the ring
        env.backend.call("action_step", a, owner=self)
        if not self._quiet_action_errors:
            self._watch_flags()
        # Set target positions (position_action_manager.py:417)
        env.robot.control_dofs_position(self._actions, self.dofs_idx)
        return self._actions

    def handle_actions(self, actions: torch.Tensor) -> torch.Tensor:
        """position_action_manager.py:389-419: actions → position targets, sent to the actuators.  A subclass that overrides it is
        called by step() with the raw actions, and reaches this one (the native launch) through ``super().handle_actions(...)``."""
        return self._process(actions, False)

    def _watch_flags(self):
        """The reference prints on NaN/Inf actions after two blocking ``.any()`` calls per step
        (position_action_manager.py:402-406); here the flag word rides along in the lazily read stats block."""
        log = self.env._extras.get(self.env.extras_logging_key)
        if hasattr(log, "add_filler"):
            log.add_filler(self._report_flags)

    @staticmethod
    def _report_flags(st, out):
        if st.action_flags & 1:
            print("ERROR: NaN actions received!")
        if st.action_flags & 2:
            print("ERROR: Infinite actions received!")

    # -- reset --------------------------------------------------------------------------------------
    def _upload_gains(self, envs_idx):
        robot = self.env.robot
        if self._kp_values is not None:
            robot.set_dofs_kp(self._add_random_noise(self._kp_values, self._noise_scale), self.dofs_idx, envs_idx)
        if self._kv_values is not None:
            robot.set_dofs_kv(self._add_random_noise(self._kv_values, self._noise_scale), self.dofs_idx, envs_idx)
        if self._damping_values is not None:
            robot.set_dofs_damping(self._add_random_noise(self._damping_values, self._noise_scale), self.dofs_idx, envs_idx)
        if self._stiffness_values is not None:
            robot.set_dofs_stiffness(self._add_random_noise(self._stiffness_values, self._noise_scale), self.dofs_idx, envs_idx)
        if self._frictionloss_values is not None:
            robot.set_dofs_frictionloss(self._add_random_noise(self._frictionloss_values, self._noise_scale), self.dofs_idx, envs_idx)
        if self._force_range is not None:
            lower = self._add_random_noise(self._force_range[0], self._noise_scale)
            upper = self._add_random_noise(self._force_range[1], self._noise_scale)
            robot.set_dofs_force_range(lower, upper, self.dofs_idx, envs_idx)

    def reset(self, envs_idx: list[int] = None):
        """position_action_manager.py:421-464 with an explicit index list (public API / real Genesis)."""
        if not self.enabled:
            return
        if envs_idx is None:
            envs_idx = torch.arange(self.env.num_envs, device=gs.device)
        self._upload_gains(envs_idx)
        position = self._add_random_noise(self._default_dofs_pos[envs_idx], self._noise_scale)
        self.env.robot.set_dofs_position(position=position, dofs_idx_local=self.dofs_idx, envs_idx=envs_idx)

    def _can_fuse_reset(self) -> bool:
        ad = self.env._adapter
        return hasattr(self.env.robot, "gf_masked_dofs") or (ad is not None and ad.setters_verified is not False)

    def _fill_reset(self, a: nat.GfResetArgs) -> None:
        """Scene-side section of the fused reset: dof_pos <- default (+noise), dof_vel <- 0.  PD gains are
        per-env constants unless ``noise_scale`` is set; they are uploaded once at the first reset."""
        robot = self.env.robot
        if not getattr(self, "_gains_uploaded", False) or self._noise_scale != 0.0:
            self._upload_gains(None)
            self._gains_uploaded = True
        if hasattr(robot, "gf_masked_dofs"):
            pos, vel = robot.gf_masked_dofs(self.dofs_idx)
        else:
            # Genesis-shaped scene: the reset rows are written into this tick's snapshot (what the getters return once the setter
            # below has run) and pushed into the simulator by index list (position_action_manager.py:455-464)
            ad = self.env._adapter
            idx = list(self.dofs_idx)
            pos, vel = ad.dofs(robot, "position", idx), ad.dofs(robot, "velocity", idx)

            def push(ids, robot=robot, ad=ad, idx=idx):
                robot.set_dofs_position(position=ad.dofs(robot, "position", idx)[ids], dofs_idx_local=idx, envs_idx=ids)

            ad.on_push("dofs", push)
        a.num_dofs = self.num_actions
        a.scene_dof_pos, a.scene_dof_vel = pos.data_ptr(), vel.data_ptr()
        a.default_dof_pos = self._k_default.data_ptr()
        a.dof_noise_scale = float(self._noise_scale)
        d = self.env.take_draws("dof_reset")
        self._keep = d
        a.dof_draws = None if d is None else d.data_ptr()

    # -- helpers ------------------------------------------------------------------------------------------
    def _per_dof(self, table: dict, start=None, fill=0.0) -> list:
        """``{joint-name pattern: value}`` → one value per controlled DOF.  Patterns are tried in dict order against the DOFs no
        earlier pattern has claimed (the first match wins); a pattern that claims nothing is an error, as in the reference
        (position_action_manager.py:470-501: ``Joint DOF '<pattern>' not found.``).  ``start``: initial values (the joint limits
        for ``clip``); sequences are stored as lists."""
        names = list(self._enabled_dof)
        out = list(start) if start is not None else [fill] * len(names)
        unclaimed = list(range(len(names)))
        for pattern, value in table.items():
            rx = re.compile(f"^{pattern}$")
            hits = [i for i in unclaimed if rx.match(names[i])]
            if not hits:
                raise RuntimeError(f"Joint DOF '{pattern}' not found.")
            for i in hits:
                out[i] = list(value) if isinstance(value, (tuple, list)) else value
            unclaimed = [i for i in unclaimed if i not in hits]
        return out

    def _get_dof_value_array(self, values, default_value=0.0, output=None):   # (the reference's private name, kept as an alias)
        return self._per_dof(values, start=output, fill=default_value)

    def _add_random_noise(self, values: torch.Tensor, noise_scale: float = 0.0) -> torch.Tensor:
        """``values + U(-1, 1) * noise_scale`` (torch's generator: gains are uploaded outside the step's kernels)."""
        if noise_scale == 0.0:
            return values
        return values + (torch.rand_like(values) * 2.0 - 1.0) * noise_scale


class PositionWithinLimitsActionManager(PositionActionManager):
    """Actions in [-1, 1] mapped onto the joint limits (position_within_limits.py:9-131):
    ``clamp(a, -1, 1) * (hi-lo)/2 + (hi+lo)/2``."""

    _mode = nat.GF_ACTION_WITHIN_LIMITS

    def __init__(self, env, joint_names=".*", default_pos={".*": 0.0}, pd_kp=None, pd_kv=None, max_force=None, damping=None,
                 stiffness=None, frictionloss=None, noise_scale: float = 0.0, action_handler=None,
                 quiet_action_errors: bool = False, delay_step: int = 0):
        super().__init__(env, joint_names=joint_names, default_pos=default_pos, pd_kp=pd_kp, pd_kv=pd_kv, max_force=max_force,
                         damping=damping, stiffness=stiffness, frictionloss=frictionloss, noise_scale=noise_scale,
                         action_handler=action_handler, quiet_action_errors=quiet_action_errors, delay_step=delay_step)

    def _build_native(self):
        super()._build_native()
        lower, upper = self.env.robot.get_dofs_limit(self.dofs_idx)
        lower, upper = lower.to(gs.tc_float), upper.to(gs.tc_float)
        self._offset = ((upper + lower) * 0.5).unsqueeze(0).expand(self.env.num_envs, -1)
        self._scale = ((upper - lower) * 0.5).unsqueeze(0).expand(self.env.num_envs, -1)
        self._k_offset = ((upper + lower) * 0.5).contiguous()
        self._k_scale = ((upper - lower) * 0.5).contiguous()
