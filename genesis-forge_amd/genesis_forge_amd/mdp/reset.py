"""
Entity reset functions — same names and arguments as genesis_forge/mdp/reset.py.

``position`` (reset.py:67-124), used by every BASELINE config except rough_terrain, is absorbed into
the fused masked reset when the scene exposes masked setters; called directly it behaves like the
reference (index-list scatter through the entity's setters).  The others call straight into the
entity's setters — there is no per-env math to fuse.
"""
from __future__ import annotations

import math
from typing import Callable, Literal

import torch

from .. import gs
from ..managers.config import ResetMdpFnClass


def zero_all_dofs_velocity(env, entity, envs_idx):
    entity.zero_all_dofs_velocity(envs_idx)


def xyz_to_quat(xyz: torch.Tensor) -> torch.Tensor:
    """Euler xyz (radians, extrinsic x-y-z) → quaternion (w,x,y,z); stands in for genesis.utils.geom.xyz_to_quat."""
    hx, hy, hz = xyz[..., 0] * 0.5, xyz[..., 1] * 0.5, xyz[..., 2] * 0.5
    cx, sx, cy, sy, cz, sz = torch.cos(hx), torch.sin(hx), torch.cos(hy), torch.sin(hy), torch.cos(hz), torch.sin(hz)
    return torch.stack([cx * cy * cz + sx * sy * sz, sx * cy * cz - cx * sy * sz, cx * sy * cz + sx * cy * sz,
                        cx * cy * sz - sx * sy * cz], dim=-1)


def set_rotation(env, entity, envs_idx, x=0, y=0, z=0):
    """Randomised euler rotation (reset.py:33-64).  As in the reference's code, only axes given as ``(lo, hi)`` tuples are
    written — a scalar angle is accepted and ignored (the angle buffer stays 0 for that axis, :52-58)."""
    angle_buffer = torch.zeros((len(envs_idx), 3), device=gs.device)
    for k, v in enumerate((x, y, z)):
        if isinstance(v, tuple):
            angle_buffer[:, k].uniform_(*v)
    entity.set_quat(xyz_to_quat(angle_buffer), envs_idx=envs_idx)


class position(ResetMdpFnClass):
    """Fixed spawn pose: ``params = {"position": (x, y, z), "quat": (w, x, y, z) | None, "zero_velocity": bool}`` — the
    ``on_reset`` entry of every shipped example but rough_terrain (reference: mdp/reset.py:67-124).

    Inside a step this object is only DESCRIBED to the native reset (``EntityManager._fill_reset`` reads ``reset_pos`` /
    ``reset_quat`` / ``zero_velocity``; the pose is written per lane by ``gf_masked_reset`` / the fused launch).  Calling it with
    an index list — the public signature — hands the pose to the entity's ``envs_idx`` setters as a broadcast view: one row
    per listed env, no per-env staging buffer to scatter into and gather back from."""

    def __init__(self, env, entity, position, quat=None, zero_velocity: bool = True):
        self.env = env
        self.zero_velocity = bool(zero_velocity)
        self.reset_pos = torch.as_tensor(position, device=gs.device, dtype=gs.tc_float).reshape(3)
        self.reset_quat = None if quat is None else torch.as_tensor(quat, device=gs.device, dtype=gs.tc_float).reshape(4)

    def __call__(self, env, entity, envs_idx, position=None, quat=None, zero_velocity: bool = True):
        ids = torch.as_tensor(envs_idx, device=gs.device)
        rows = int(ids.numel())
        if rows == 0:
            return
        entity.set_pos(self.reset_pos.expand(rows, 3), envs_idx=envs_idx, zero_velocity=self.zero_velocity)
        if self.reset_quat is not None:
            entity.set_quat(self.reset_quat.expand(rows, 4), envs_idx=envs_idx, zero_velocity=self.zero_velocity)


class randomize_terrain_position(ResetMdpFnClass):
    """Random position on the terrain and (by default) a random yaw (reset.py:127-226).

    On an entity with masked setters the EntityManager folds this whole function into the masked reset (``GfResetArgs.spawn_*``:
    the draws, the terrain-height lookup, ``xyz_to_quat`` and the pose write happen per lane, no ``nonzero`` / gather / scatter);
    ``gf_spawn`` describes it.  Called directly it runs the reference's sequence through the public setters."""

    def __init__(self, env, entity, terrain_manager, height_offset: float = 0.1e-3, subterrain=None,
                 rotation: dict | None = {"z": (0, 2 * math.pi)}, zero_velocity: bool = True):
        self.env = env
        self.rotation = rotation
        self._rotation_buffer = None
        self._quat_buffer = None

    def build(self):
        self._rotation_buffer = torch.zeros((self.env.num_envs, 3), device=gs.device, dtype=gs.tc_float)
        self._quat_buffer = torch.zeros((self.env.num_envs, 4), device=gs.device, dtype=gs.tc_float)

    def define_quat(self, envs_idx, rotation: dict):
        """Only axes given as ``(lo, hi)`` tuples are written; scalars leave the axis at 0 (reset.py:172-196)."""
        for k, axis in enumerate("xyz"):
            v = rotation.get(axis, 0)
            if isinstance(v, tuple):
                self._rotation_buffer[envs_idx, k] = torch.empty(len(envs_idx), device=gs.device).uniform_(*v)
        self._quat_buffer[envs_idx] = xyz_to_quat(self._rotation_buffer[envs_idx])

    def gf_spawn(self, terrain_manager, height_offset: float = 0.1e-3, subterrain=None, rotation: dict | None = {"z": (0, 2 * math.pi)},
                 zero_velocity: bool = True):
        """(usable area, height offset, rotation ranges, zero_velocity) for the fused reset, or None when it cannot be described
        statically (``subterrain`` given as a callable is re-evaluated by the reference on every reset)."""
        if callable(subterrain) or not hasattr(terrain_manager, "gf_view"):
            return None
        rot = None
        if rotation is not None:
            rot = [rotation.get(axis) if isinstance(rotation.get(axis), tuple) else None for axis in "xyz"]
        return terrain_manager.usable_area(0.5, subterrain), float(height_offset), rot, bool(zero_velocity)

    def __call__(self, env, entity, envs_idx, terrain_manager, height_offset: float = 0.1e-3, subterrain=None,
                 rotation: dict | None = {"z": (0, 2 * math.pi)}, zero_velocity: bool = True):
        sub = subterrain() if callable(subterrain) else subterrain
        pos = terrain_manager.generate_random_env_pos(envs_idx=envs_idx, subterrain=sub, height_offset=height_offset)
        entity.set_pos(pos, envs_idx=envs_idx, zero_velocity=zero_velocity)
        if rotation is not None:
            self.define_quat(envs_idx, rotation)
            entity.set_quat(self._quat_buffer[envs_idx], envs_idx=envs_idx, zero_velocity=zero_velocity)


class randomize_link_mass_shift(ResetMdpFnClass):
    """Mass shift of the links whose name matches ``link_name`` (reference: mdp/reset.py:229-284) — a pure call into Genesis'
    ``set_mass_shift`` at reset time, nothing per env to compute (out of the hot path).

    What reaches the simulator is pinned to the reference's OBSERVABLE behaviour rather than to its intent: there the uniform
    draw lands in the temporary that advanced indexing returns (:271), so the shift handed to ``set_mass_shift`` is the all-zero
    ``[num_envs, n_links]`` tensor, passed whole together with ``envs_idx`` (:274-278).  Here that is stated directly: a zero
    shift per matching link, one draw of the same shape taken from torch's generator so the global RNG stream advances exactly
    as it does in the reference."""

    def __init__(self, _env, entity, link_name: str, add_mass_range: tuple[float, float] = (-0.2, 0.2)):
        self.env = _env
        self.add_mass_range = add_mass_range
        self._entity = entity
        self._link_name = link_name
        self.build()

    def build(self):
        from ..utils import links_by_name_pattern

        matches = links_by_name_pattern(self._entity, self._link_name) if self._link_name is not None else []
        self._links_idx_local = [link.idx_local for link in matches]
        self._zero_shift = torch.zeros((self.env.num_envs, len(self._links_idx_local)), device=gs.device) if matches else None

    def __call__(self, env, entity, envs_idx, link_name: str, add_mass_range: tuple[float, float] = (-0.2, 0.2)):
        lo, hi = self.add_mass_range
        torch.empty((len(envs_idx), len(self._links_idx_local)), device=gs.device).uniform_(lo, hi)   # (drawn and dropped, as upstream)
        self._entity.set_mass_shift(self._zero_shift, links_idx_local=self._links_idx_local, envs_idx=envs_idx)
