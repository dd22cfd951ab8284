"""
EntityManager — API mirror of genesis_forge/managers/entity_manager.py.

The reference caches ``base_pos`` / ``base_quat`` / ``inv_base_quat`` each step (:189-195) and every
term that asks for a body-frame vector re-runs ``transform_by_quat`` (:130-146, ~7-10 launches each).
Here the fused phase kernels compute those rotations in registers from the raw quaternion, so ``step``
only refreshes the per-tick view cache; the public getters stay available (one ``gf_entity_rotate``
launch each) for opaque user terms and carry a provenance tag so ObservationManager can fuse lambdas
that merely forward them.  ``reset`` runs the ``on_reset`` ConfigItems (:169-183); the fixed-pose
``mdp.reset.position`` is absorbed into the fused masked reset when the scene exposes masked setters.
"""
from __future__ import annotations

from typing import Any, Optional

import torch

from .. import _native as nat
from .. import gs
from .action import _tag
from .base import BaseManager
from .config import ConfigItem


class EntityManager(BaseManager):
    """Manages one scene entity's reset and shared body-frame quantities (ctor as entity_manager.py:80-98)."""

    _fused_reset = True

    def __init__(self, env, entity_attr: str, on_reset: dict[str, dict] | None = None):
        super().__init__(env, "entity")
        self.entity = None
        self._entity_attr = entity_attr
        self.on_reset: dict[str, ConfigItem] = {}
        for name, cfg in (on_reset or {}).items():
            self.on_reset[name] = ConfigItem(cfg, env, on_dirty=getattr(env, "invalidate_trace", None))  # params feed the fused reset
        N = env.num_envs
        self._global_gravity = torch.tensor([0.0, 0.0, -1.0], device=gs.device, dtype=gs.tc_float).expand(N, 3)
        self._base_pos = torch.zeros(N, 3, device=gs.device, dtype=gs.tc_float)
        self._base_quat = torch.zeros(N, 4, device=gs.device, dtype=gs.tc_float)
        self._rot_args = nat.GfRotateArgs()
        # pre-reset quaternions of the envs reset in the current tick (see _stale_views)
        self._stash = torch.zeros(N, 4, device=gs.device, dtype=gs.tc_float)
        self._stale_masks = None
        self._stale_tick = -1
        self._stash_armed = False

    # -- properties -----------------------------------------------------------------------------------
    def stale(self):
        """(stash, mask, mask2) while the quirk of entity_manager.py:189-195 applies: the reference caches
        ``base_quat`` in ``step()`` and does not refresh it after the reset later in the same tick, so until the next
        ``step()`` the body-frame getters rotate just-reset envs by their PRE-reset orientation."""
        if self._stale_masks is not None and self._stale_tick == self.env._tick:
            return (self._stash,) + self._stale_masks
        return None

    def _views(self):
        v = self.env.entity_views(self.entity)
        st = self.stale()
        if st is None:
            return v
        stash, m1, m2 = st
        m = m1 if m2 is None else (m1 | m2)
        from ..genesis_env import EntityViews
        return EntityViews(v.pos, torch.where(m.unsqueeze(-1), stash, v.quat), v.lin_vel, v.ang_vel)

    @property
    def base_pos(self) -> torch.Tensor:
        return self._views().pos

    @property
    def base_quat(self) -> torch.Tensor:
        return self._views().quat

    @property
    def inv_base_quat(self) -> torch.Tensor:
        q = self._views().quat.clone()  # inv_quat = conjugate (entity_manager.py:195)
        q[:, 1:] *= -1
        return q

    # -- helpers (entity_manager.py:130-146) ----------------------------------------------------------
    def _rotate(self, what: int, tag: str) -> torch.Tensor:
        env = self.env
        out = torch.empty(env.num_envs, 3, device=gs.device, dtype=gs.tc_float)
        a = self._rot_args
        a.num_envs, a.what = env.num_envs, what
        v = self._views()
        v.fill(a.entity)
        a.out = out.data_ptr()
        env.backend.call("entity_rotate", a, owner=None)
        self._keep_views = v  # the stale-quat composition is a temporary: keep it alive past the launch
        return _tag(out, (tag, self))

    def get_projected_gravity(self) -> torch.Tensor:
        return self._rotate(nat.GF_ROT_PROJ_GRAVITY, "grav")

    def get_linear_velocity(self) -> torch.Tensor:
        return self._rotate(nat.GF_ROT_LIN_VEL, "lin_vel")

    def get_angular_velocity(self) -> torch.Tensor:
        return self._rotate(nat.GF_ROT_ANG_VEL, "ang_vel")

    # -- operations -----------------------------------------------------------------------------------
    def build(self):
        self.entity = getattr(self.env, self._entity_attr)
        for cfg in self.on_reset.values():
            cfg.build(entity=self.entity)

    def step(self):
        """The reference re-fetches pos/quat here (:163-167); the per-tick view cache does that lazily."""

    def reset(self, envs_idx: list[int] | None = None):
        """Run every on_reset fn for ``envs_idx`` (entity_manager.py:169-183)."""
        if not self.enabled:
            return
        if envs_idx is None:
            envs_idx = torch.arange(self.env.num_envs, device=gs.device)
        for name, cfg in self.on_reset.items():
            try:
                cfg.execute(envs_idx)
            except Exception as e:
                print(f"Error resetting entity with config: '{name}'")
                raise e
        self.env.invalidate_views()

    # -- fused reset --------------------------------------------------------------------------------
    def _can_fuse_reset(self) -> bool:
        """True when the scene has masked setters and the on_reset entry is a fixed-pose ``mdp.reset.position`` or a
        ``mdp.reset.randomize_terrain_position`` whose arguments are static (see its ``gf_spawn``)."""
        ad = self.env._adapter
        if not hasattr(self.entity, "gf_masked_base") and (ad is None or ad.setters_verified is False):
            return False
        from ..mdp import reset as reset_mdp
        items = list(self.on_reset.values())
        if len(items) > 1:
            return False
        for c in items:
            if isinstance(c.fn, reset_mdp.randomize_terrain_position):
                if c.fn.gf_spawn(**c.params) is None:
                    return False
            elif not isinstance(c.fn, reset_mdp.position):
                return False
        return True

    def _after_fused_reset(self, mask, mask2) -> None:
        if getattr(self, "_stash_armed", False) or getattr(self, "_stash_always", False):
            self._stale_masks = (mask, mask2)
            self._stale_tick = self.env._tick + 1  # the views cache is invalidated right after the reset
            self._stash_always = self._stash_always if hasattr(self, "_stash_always") else False
            if self._stash_armed:
                self._stash_always = True  # a recorded step replays the same descriptor (quat_stash stays set)
            self._stash_armed = False

    def _masked_base(self, cfg):
        """The base-state tensors the masked reset writes.  Synthetic scene: the scene's own buffers.  Genesis-shaped scene: this
        tick's snapshot, plus the write-back of the reset rows through ``set_pos`` / ``set_quat`` by index list, in the order
        and with the ``zero_velocity`` argument of mdp/reset.py:102-124 / :213-226."""
        ent = self.entity
        if hasattr(ent, "gf_masked_base"):
            return ent.gf_masked_base()
        env = self.env
        v = env.entity_views(ent)

        def push(ids, self=self, cfg=cfg, ent=ent, env=env):
            from ..mdp import reset as reset_mdp
            fn = cfg.fn
            v = env.entity_views(ent)
            if isinstance(fn, reset_mdp.randomize_terrain_position):
                _area, _off, rot, zero_velocity = fn.gf_spawn(**cfg.params)
                with_quat = rot is not None
            else:
                zero_velocity, with_quat = fn.zero_velocity, fn.reset_quat is not None
            ent.set_pos(v.pos[ids], envs_idx=ids, zero_velocity=zero_velocity)
            if with_quat:
                ent.set_quat(v.quat[ids], envs_idx=ids, zero_velocity=zero_velocity)

        env._adapter.on_push("base:%d" % id(ent), push)
        return v.pos, v.quat, v.lin_vel, v.ang_vel

    def _fill_reset(self, a: nat.GfResetArgs) -> None:
        from ..mdp import reset as reset_mdp
        for cfg in self.on_reset.values():
            fn = cfg.fn
            pos, quat, lin, ang = self._masked_base(cfg)
            a.scene_pos, a.scene_quat = pos.data_ptr(), quat.data_ptr()
            a.scene_lin_vel, a.scene_ang_vel = lin.data_ptr(), ang.data_ptr()
            if isinstance(fn, reset_mdp.randomize_terrain_position):
                # re-read every reset: params (height_offset, rotation ranges) may be mutated like any other cfg entry
                (x_min, x_max, y_min, y_max), offset, rot, zero_velocity = fn.gf_spawn(**cfg.params)
                a.spawn_mode = 1
                a.spawn_x_min, a.spawn_x_span = x_min, x_max - x_min   # rand * (x_max - x_min) + x_min, terrain_manager.py:236-241
                a.spawn_y_min, a.spawn_y_span = y_min, y_max - y_min
                a.spawn_height_offset = offset
                a.spawn_set_quat, a.spawn_rot_mask = (0 if rot is None else 1), 0
                for k in range(3):
                    if rot is not None and rot[k] is not None:
                        a.spawn_rot_mask |= 1 << k
                        a.spawn_rot_lo[k], a.spawn_rot_hi[k] = float(rot[k][0]), float(rot[k][1])
                cfg.params["terrain_manager"].gf_view(a.terrain)
                d = self.env.take_draws("spawn")
                self._keep_spawn = d
                a.spawn_draws = None if d is None else d.data_ptr()
                a.zero_velocity = 1 if zero_velocity else 0
                if rot is not None:
                    a.quat_stash = self._stash.data_ptr()
                    self._stash_armed = True
                continue
            host = getattr(fn, "_gf_host_pose", None)
            if host is None:  # read the fixed pose back once; never per step (a device->host copy is a sync)
                host = (fn.reset_pos.tolist(), None if fn.reset_quat is None else fn.reset_quat.tolist())
                fn._gf_host_pose = host
            for j in range(3):
                a.reset_pos[j] = host[0][j]
            a.set_quat = 0
            if host[1] is not None:
                a.set_quat = 1
                a.quat_stash = self._stash.data_ptr()
                self._stash_armed = True
                for j in range(4):
                    a.reset_quat[j] = host[1][j]
            a.zero_velocity = 1 if fn.zero_velocity else 0


# -- annotation type of the reference (entity_manager.py:17-40) -----------------------------------------------------------------------
from typing import Callable, TypedDict  # noqa: E402


class EntityResetConfig(TypedDict, total=False):
    """One ``on_reset`` entry: ``fn(env, entity, envs_idx, **params)`` — a function or a ResetMdpFnClass — and its ``params``."""
    fn: Callable
    params: dict
