"""
Reward terms — same names, arguments and results as genesis_forge/mdp/rewards.py.

Each function is *two* things:
  * a descriptor: ``fn._gf_spec(env, **params)`` tells RewardManager which opcode of the fused
    ``gf_reward_step`` kernel implements it and which buffers it reads, so a cfg built from these
    functions runs as one launch;
  * a callable with the reference's signature: calling it directly (inside a user lambda, in a test)
    evaluates that single term on device through the same kernel (EVAL mode) and returns ``[N]`` f32.
"""
from __future__ import annotations

import math
from typing import Union

import torch

from .. import _native as nat
from .. import gs
from ..managers._program import TermSpec, eval_reward_spec
from ..managers.config import MdpFnClass


def _entity(env, entity_attr, entity_manager):
    return entity_manager.entity if entity_manager is not None else getattr(env, entity_attr)


def _term(spec_fn):
    """Build the public function from its spec maker."""

    def deco(fn):
        def public(env, *args, **kwargs):
            spec = spec_fn(env, *args, **kwargs)
            if spec is None:  # no opcode for this combination of arguments: the function body is the torch restatement
                return fn(env, *args, **kwargs)
            return eval_reward_spec(env, spec)

        public.__name__ = fn.__name__
        public.__qualname__ = fn.__qualname__
        public.__doc__ = fn.__doc__
        public._gf_spec = spec_fn
        return public

    return deco


# -- aliveness (rewards.py:31-46) -------------------------------------------------------------------
@_term(lambda env: TermSpec(nat.GF_R_IS_ALIVE, needs_terminated=True))
def is_alive(env):
    """1 for envs that did not terminate this step (reads ``env.extras["terminations"]``)."""


@_term(lambda env: TermSpec(nat.GF_R_TERMINATED, needs_terminated=True))
def terminated(env):
    """1 for envs that terminated this step."""


# -- base state (rewards.py:54-193) -----------------------------------------------------------------
def _spec_base_height(env, target_height: Union[float, torch.Tensor] = None, height_command=None, terrain_manager=None,
                      entity_attr: str = "robot", entity_manager=None):
    ent = _entity(env, entity_attr, entity_manager)
    flags, cmd, ext, p0, terrain = 0, {}, {}, 0.0, None
    if height_command is not None:
        flags |= nat.GF_RW_FLAG_CMD
        cmd[0] = height_command
    elif isinstance(target_height, torch.Tensor):
        flags |= nat.GF_RW_FLAG_CMD
        cmd[0] = target_height
    else:
        p0 = float(target_height)
    if terrain_manager is not None:
        if not hasattr(terrain_manager, "gf_view"):
            return None  # a foreign terrain object: the whole term is evaluated by Python (EXTERNAL column)
        flags |= nat.GF_RW_FLAG_TERRAIN  # the kernel samples the manager's height field under the base
        terrain = terrain_manager
    return TermSpec(nat.GF_R_BASE_HEIGHT, p=[p0], flags=flags, entity=ent, cmd=cmd, ext=ext, terrain=terrain)


@_term(_spec_base_height)
def base_height(env, target_height=None, height_command=None, terrain_manager=None, entity_attr="robot", entity_manager=None):
    """``(base_z - terrain_height - target)^2`` (rewards.py:54-90)."""
    # only reached with a terrain object that is not this package's TerrainManager
    pos = _entity(env, entity_attr, entity_manager).get_pos()
    offset = terrain_manager.get_terrain_height(pos[:, 0], pos[:, 1]) if terrain_manager is not None else 0.0
    if height_command is not None:
        target_height = height_command.command.squeeze(-1)
    return torch.square(pos[:, 2] - offset - target_height)


def _spec_dof_similar(env, action_manager):
    return TermSpec(nat.GF_R_DOF_SIMILAR_TO_DEFAULT, action_manager=action_manager)


@_term(_spec_dof_similar)
def dof_similar_to_default(env, action_manager):
    """``sum_d |dof_pos - default_pos|`` (rewards.py:93-109)."""


@_term(lambda env, entity_attr="robot", entity_manager=None: TermSpec(nat.GF_R_LIN_VEL_Z_L2, entity=_entity(env, entity_attr, entity_manager)))
def lin_vel_z_l2(env, entity_attr="robot", entity_manager=None):
    """Squared z component of the body-frame linear velocity (rewards.py:112-135)."""


@_term(lambda env, entity_attr="robot", entity_manager=None: TermSpec(nat.GF_R_ANG_VEL_XY_L2, entity=_entity(env, entity_attr, entity_manager)))
def ang_vel_xy_l2(env, entity_attr="robot", entity_manager=None):
    """Sum of squared x/y body-frame angular velocity (rewards.py:138-161)."""


@_term(lambda env, entity_attr="robot", entity_manager=None: TermSpec(nat.GF_R_FLAT_ORIENTATION_L2, entity=_entity(env, entity_attr, entity_manager)))
def flat_orientation_l2(env, entity_attr="robot", entity_manager=None):
    """Sum of squared x/y projected gravity (rewards.py:164-193)."""


class body_acceleration_exp(MdpFnClass):
    """``1 - exp(-sensitivity * (|d v_lin/dt| + |d v_ang/dt|))`` with per-instance previous velocities
    (rewards.py:196-249).  As in the reference the first call sees zero acceleration and the previous
    velocities are not cleared on env reset."""

    def __init__(self, env, entity_attr: str = "robot", entity_manager=None, sensitivity: float = 0.10):
        super().__init__(env)
        self._entity_attr = entity_attr
        self._state = torch.zeros(env.num_envs, 6, device=gs.device, dtype=gs.tc_float)
        self._called = False

    def _mark_called(self):
        self._called = True

    def _gf_spec(self, env, entity_attr: str = "robot", entity_manager=None, sensitivity: float = 0.10):
        ent = _entity(env, entity_attr, entity_manager)
        flags = 0 if self._called else nat.GF_RW_FLAG_FIRST_CALL
        return TermSpec(nat.GF_R_BODY_ACCEL_EXP, p=[float(sensitivity)], flags=flags, entity=ent, state={0: self._state},
                        after=None if self._called else self._mark_called)

    @property
    def prev_lin_vel(self):
        return self._state[:, :3]

    @property
    def prev_ang_vel(self):
        return self._state[:, 3:]

    def __call__(self, env, entity_attr: str = "robot", entity_manager=None, sensitivity: float = 0.10):
        return eval_reward_spec(env, self._gf_spec(env, entity_attr, entity_manager, sensitivity))


# -- action penalties (rewards.py:257-271) ----------------------------------------------------------
@_term(lambda env: TermSpec(nat.GF_R_ACTION_RATE_L2, needs_actions=True))
def action_rate_l2(env):
    """``sum_d (last_actions - actions)^2`` on the raw policy actions."""


# -- velocity command tracking (rewards.py:279-385) ---------------------------------------------------
def _spec_track_lin(env, command: torch.Tensor = None, vel_cmd_manager=None, sensitivity: float = 0.25, entity_attr="robot",
                    entity_manager=None):
    assert command is not None or vel_cmd_manager is not None, "Either command or vel_cmd_manager must be provided to command_tracking_lin_vel"
    src = vel_cmd_manager if vel_cmd_manager is not None else command
    return TermSpec(nat.GF_R_CMD_TRACK_LIN_VEL, p=[float(sensitivity)], entity=_entity(env, entity_attr, entity_manager), cmd={0: src})


@_term(_spec_track_lin)
def command_tracking_lin_vel(env, command=None, vel_cmd_manager=None, sensitivity=0.25, entity_attr="robot", entity_manager=None):
    """``exp(-|cmd_xy - v_xy|^2 / sensitivity)`` (rewards.py:279-317)."""


def _spec_track_ang(env, commanded_ang_vel: torch.Tensor = None, vel_cmd_manager=None, sensitivity: float = 0.25, entity_attr="robot",
                    entity_manager=None):
    assert commanded_ang_vel is not None or vel_cmd_manager is not None, "Either commanded_ang_vel or vel_cmd_manager must be provided to command_tracking_ang_vel"
    if vel_cmd_manager is not None:
        src, col = vel_cmd_manager, 2
    else:
        src, col = commanded_ang_vel, 0
    return TermSpec(nat.GF_R_CMD_TRACK_ANG_VEL, p=[float(sensitivity)], i=[0, col], entity=_entity(env, entity_attr, entity_manager), cmd={0: src})


@_term(_spec_track_ang)
def command_tracking_ang_vel(env, commanded_ang_vel=None, vel_cmd_manager=None, sensitivity=0.25, entity_attr="robot", entity_manager=None):
    """``exp(-(cmd_z - w_z)^2 / sensitivity)`` (rewards.py:320-358)."""


def _spec_stand_still(env, command_threshold: float = 0.06, vel_cmd_manager=None, action_manager=None):
    return TermSpec(nat.GF_R_STAND_STILL, p=[float(command_threshold)], action_manager=action_manager, cmd={0: vel_cmd_manager})


@_term(_spec_stand_still)
def stand_still_joint_deviation_l1(env, command_threshold=0.06, vel_cmd_manager=None, action_manager=None):
    """Joint deviation from default, only while the xy command is below the threshold (rewards.py:361-385)."""


# -- contacts (rewards.py:393-504) --------------------------------------------------------------------
def _spec_has_contact(_env, contact_manager, threshold=1.0, min_contacts=1):
    return TermSpec(nat.GF_R_HAS_CONTACT, p=[float(threshold)], i=[0, int(min_contacts)], contact={0: contact_manager})


@_term(_spec_has_contact)
def has_contact(_env, contact_manager, threshold=1.0, min_contacts=1):
    """1 when at least ``min_contacts`` tracked links feel more than ``threshold`` N (rewards.py:393-410)."""


def _spec_contact_force(_env, contact_manager, threshold: float = 1.0):
    return TermSpec(nat.GF_R_CONTACT_FORCE, p=[float(threshold)], contact={0: contact_manager})


@_term(_spec_contact_force)
def contact_force(_env, contact_manager, threshold: float = 1.0):
    """Total force above the threshold over the tracked links (rewards.py:413-428)."""


def _spec_feet_air_time(env, contact_manager, time_threshold: float, time_threshold_max: float | None = None, vel_cmd_manager=None):
    flags, p1 = 0, 0.0
    if time_threshold_max is not None:
        flags |= nat.GF_RW_FLAG_MAX
        p1 = time_threshold_max - time_threshold
    return TermSpec(nat.GF_R_FEET_AIR_TIME, p=[float(time_threshold), float(p1), env.dt + 1.0e-8], flags=flags,
                    contact={0: contact_manager}, cmd={1: vel_cmd_manager})


@_term(_spec_feet_air_time)
def feet_air_time(env, contact_manager, time_threshold, time_threshold_max=None, vel_cmd_manager=None):
    """Rewards long steps: air time above a threshold on the step a foot lands (rewards.py:431-469)."""


def _spec_feet_slide(env, contact_manager, entity_attr: str = "robot"):
    return TermSpec(nat.GF_R_FEET_SLIDE, contact={0: contact_manager}, link_vel=True)


@_term(_spec_feet_slide)
def feet_slide(env, contact_manager, entity_attr="robot"):
    """Foot speed while in contact (rewards.py:472-504)."""
