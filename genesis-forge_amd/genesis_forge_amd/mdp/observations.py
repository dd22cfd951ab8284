"""
Observation getters — same names and arguments as genesis_forge/mdp/observations.py.  They forward to
the managers' getters, whose results carry the provenance tags ObservationManager uses to fuse them.
"""
from __future__ import annotations

import torch

from .. import _native as nat
from .. import gs
from ..managers.action import _tag
from ..utils import entity_ang_vel, entity_lin_vel
from ..utils import entity_projected_gravity as _entity_projected_gravity


def entity_linear_velocity(env, entity_manager=None, entity_attr: str = "robot") -> torch.Tensor:
    if entity_manager is not None:
        return entity_manager.get_linear_velocity()
    return entity_lin_vel(getattr(env, entity_attr))


def entity_angular_velocity(env, entity_manager=None, entity_attr: str = "robot") -> torch.Tensor:
    if entity_manager is not None:
        return entity_manager.get_angular_velocity()
    return entity_ang_vel(getattr(env, entity_attr))


def entity_projected_gravity(env, entity_manager=None, entity_attr: str = "robot") -> torch.Tensor:
    """(The reference shadows the imported helper here and would recurse without an entity_manager,
    observations.py:9,58-76; this version calls the helper.)"""
    if entity_manager is not None:
        return entity_manager.get_projected_gravity()
    return _entity_projected_gravity(getattr(env, entity_attr))


def entity_dofs_position(env, action_manager=None, entity_attr: str = "robot", dofs_idx: list[int] = None) -> torch.Tensor:
    if action_manager is not None:
        return action_manager.get_dofs_position()
    return getattr(env, entity_attr).get_dofs_position(dofs_idx)


def entity_dofs_velocity(env, action_manager=None, entity_attr: str = "robot", dofs_idx: list[int] = None) -> torch.Tensor:
    if action_manager is not None:
        return action_manager.get_dofs_velocity()
    return getattr(env, entity_attr).get_dofs_velocity(dofs_idx)


def entity_dofs_force(env, action_manager=None, entity_attr: str = "robot", dofs_idx: list[int] = None,
                      clip_to_max_force: bool = False) -> torch.Tensor:
    if action_manager is not None:
        return action_manager.get_dofs_force(clip_to_max_force=clip_to_max_force)
    return getattr(env, entity_attr).get_dofs_force(dofs_idx)


def current_actions(env, action_manager=None) -> torch.Tensor:
    if action_manager is not None:
        return action_manager.get_actions()
    if env.actions is None and env.action_space is not None:
        # called before the first reset (ObservationManager.build's trial observation): the reference would hand
        # torch.cat a None here; allocate the buffers reset() would create
        env._actions = torch.zeros((env.num_envs, env.action_space.shape[0]), device=gs.device, dtype=gs.tc_float)
        env._last_actions = torch.zeros_like(env._actions)
    return _tag(env.actions, ("raw_actions", env))


def contact_force(env, contact_manager) -> torch.Tensor:
    """Per-link contact force magnitude, shape (num_envs, num_links) (observations.py:182-193)."""
    out = torch.norm(contact_manager.contacts[:, :, :], dim=-1)
    return _tag(out, ("contact_norm", contact_manager))
