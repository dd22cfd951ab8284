"""
Entity helpers — same names as genesis_forge/utils.py:13-73.  The three body-frame vectors are one
``gf_entity_rotate`` launch each (the reference: ``inv_quat`` + ``transform_by_quat``, 7-10 launches).
"""
from __future__ import annotations

import re

import torch

from . import _native as nat
from . import gs


def _rotate(entity, what: int) -> torch.Tensor:
    from .genesis_env import _f32c
    pos, quat = _f32c(entity.get_pos()), _f32c(entity.get_quat())
    lin = _f32c(entity.get_vel()) if what == nat.GF_ROT_LIN_VEL else None
    ang = _f32c(entity.get_ang()) if what == nat.GF_ROT_ANG_VEL else None
    n = quat.shape[0]
    out = torch.empty(n, 3, device=quat.device, dtype=torch.float32)
    a = nat.GfRotateArgs()
    a.num_envs, a.what = n, what
    a.entity.pos, a.entity.quat = pos.data_ptr(), quat.data_ptr()
    a.entity.lin_vel = None if lin is None else lin.data_ptr()
    a.entity.ang_vel = None if ang is None else ang.data_ptr()
    a.out = out.data_ptr()
    nat.get_backend().call("entity_rotate", a)
    return out


def entity_lin_vel(entity) -> torch.Tensor:
    """Linear velocity in the entity's local frame (utils.py:13-24)."""
    return _rotate(entity, nat.GF_ROT_LIN_VEL)


def entity_ang_vel(entity) -> torch.Tensor:
    """Angular velocity in the entity's local frame (utils.py:27-38)."""
    return _rotate(entity, nat.GF_ROT_ANG_VEL)


def entity_projected_gravity(entity) -> torch.Tensor:
    """Projected gravity in the entity's local frame (utils.py:41-55)."""
    return _rotate(entity, nat.GF_ROT_PROJ_GRAVITY)


def links_by_name_pattern(entity, name_pattern: str) -> list:
    """Entity links whose name equals or fully matches the regex (utils.py:58-73)."""
    return [link for link in entity.links if link.name == name_pattern or re.match(f"^{name_pattern}$", link.name)]
