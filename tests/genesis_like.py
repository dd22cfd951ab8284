"""
Test double: a scene with ONLY the public surface of Genesis that the manager stack calls (SURVEY.md §8b "What it calls";
reference call sites: genesis_forge/managers/entity_manager.py:163-167,189-195, managers/action/position_action_manager.py:
243-289,417,421-464, managers/contact/contact_manager.py:384-432, mdp/reset.py:102-124, managed_env.py:292,308-310).

* every getter returns a NEW tensor (a copy of the simulator's state) — writing into it changes nothing in the simulator;
* state is changed through ``envs_idx`` setters only; ``control_dofs_position`` COPIES the targets at call time;
* contacts come from ``rigid_solver.collider.get_contacts(as_tensor=True, to_torch=True)`` (a dict of new tensors), link
  orientations from ``rigid_solver.get_links_quat()``;
* there is no ``gf_*`` attribute anywhere: the package must take the paths it would take on real Genesis.

The physics behind it is the package's synthetic stand-in (``genesis_forge_amd.scene.SyntheticScene``), kept PRIVATE: the same
seeds give the same trajectories as an env built directly on the synthetic scene, which is what the parity tests compare.
Two test aids: ``calls`` counts every public call; with ``poison=True`` a tensor handed out by a getter is overwritten with NaN
two ticks later — whoever still reads it (a pointer frozen into a recorded step) produces NaNs.
"""
from __future__ import annotations

import collections

import torch

from genesis_forge_amd import _native as nat
from genesis_forge_amd import scene as _synth


class _Link:
    def __init__(self, inner, entity):
        self.name, self.idx, self.idx_local = inner.name, inner.idx, inner.idx_local
        self._entity = entity

    def get_vel(self, envs_idx=None):
        return self._entity.get_links_vel(links_idx_local=[self.idx_local])[:, 0]

    def get_pos(self, envs_idx=None):
        return self._entity.get_links_pos(links_idx_local=[self.idx_local])[:, 0]


class GenesisLikeEntity:
    """``RigidEntity`` look-alike over a private synthetic entity."""

    def __init__(self, scene, inner):
        self._scene = scene
        self._e = inner
        self.joints = list(inner.joints)
        self.links = [_Link(l, self) for l in inner.links]
        self.n_links = inner.n_links
        self.n_dofs = inner.n_dofs

    # -- getters: new tensors ---------------------------------------------------------------------------------------------
    def _out(self, name: str, t: torch.Tensor) -> torch.Tensor:
        return self._scene._hand_out(name, t)

    def get_pos(self, envs_idx=None):
        return self._out("get_pos", self._e.get_pos(envs_idx))

    def get_quat(self, envs_idx=None):
        return self._out("get_quat", self._e.get_quat(envs_idx))

    def get_vel(self, envs_idx=None):
        return self._out("get_vel", self._e.get_vel(envs_idx))

    def get_ang(self, envs_idx=None):
        return self._out("get_ang", self._e.get_ang(envs_idx))

    def get_dofs_position(self, dofs_idx_local=None, envs_idx=None):
        return self._out("get_dofs_position", self._e.get_dofs_position(dofs_idx_local, envs_idx))

    def get_dofs_velocity(self, dofs_idx_local=None, envs_idx=None):
        return self._out("get_dofs_velocity", self._e.get_dofs_velocity(dofs_idx_local, envs_idx))

    def get_dofs_force(self, dofs_idx_local=None, envs_idx=None):
        return self._out("get_dofs_force", self._e.get_dofs_force(dofs_idx_local, envs_idx))

    def get_dofs_limit(self, dofs_idx_local=None):
        lo, hi = self._e.get_dofs_limit(dofs_idx_local)
        return lo.clone(), hi.clone()

    def get_links_vel(self, links_idx_local=None, envs_idx=None):
        return self._out("get_links_vel", self._e.get_links_vel(links_idx_local, envs_idx))

    def get_links_pos(self, links_idx_local=None, envs_idx=None):
        return self._out("get_links_pos", self._e.get_links_pos(links_idx_local, envs_idx))

    def get_link(self, name: str):
        for l in self.links:
            if l.name == name:
                return l
        raise KeyError(name)

    def get_AABB(self):
        return self._e.get_AABB()

    # -- control and setters ------------------------------------------------------------------------------------------------
    def control_dofs_position(self, position, dofs_idx_local=None, envs_idx=None):
        self._scene.calls["control_dofs_position"] += 1
        e = self._e
        cols = e._cols(dofs_idx_local)
        if cols is None:
            e._own_targets[:] = position          # a COPY, taken now: later edits of `position` do not reach the simulator
        else:
            e._own_targets[:, cols] = position
        e._targets = e._own_targets

    def _set(self, name):
        self._scene.calls[name] += 1

    def set_dofs_kp(self, kp, dofs_idx_local=None, envs_idx=None):
        self._set("set_dofs_kp"); self._e.set_dofs_kp(kp, dofs_idx_local, envs_idx)

    def set_dofs_kv(self, kv, dofs_idx_local=None, envs_idx=None):
        self._set("set_dofs_kv"); self._e.set_dofs_kv(kv, dofs_idx_local, envs_idx)

    def set_dofs_damping(self, v, dofs_idx_local=None, envs_idx=None):
        self._set("set_dofs_damping"); self._e.set_dofs_damping(v, dofs_idx_local, envs_idx)

    def set_dofs_stiffness(self, v, dofs_idx_local=None, envs_idx=None):
        self._set("set_dofs_stiffness"); self._e.set_dofs_stiffness(v, dofs_idx_local, envs_idx)

    def set_dofs_frictionloss(self, v, dofs_idx_local=None, envs_idx=None):
        self._set("set_dofs_frictionloss"); self._e.set_dofs_frictionloss(v, dofs_idx_local, envs_idx)

    def set_dofs_force_range(self, lower, upper, dofs_idx_local=None, envs_idx=None):
        self._set("set_dofs_force_range"); self._e.set_dofs_force_range(lower, upper, dofs_idx_local, envs_idx)

    def set_mass_shift(self, shift, links_idx_local=None, envs_idx=None):
        self._set("set_mass_shift"); self._e.set_mass_shift(shift, links_idx_local, envs_idx)

    def set_dofs_position(self, position, dofs_idx_local=None, envs_idx=None, zero_velocity: bool = True):
        self._set("set_dofs_position"); self._e.set_dofs_position(position, dofs_idx_local, envs_idx, zero_velocity)

    def zero_all_dofs_velocity(self, envs_idx=None):
        self._set("zero_all_dofs_velocity"); self._e.zero_all_dofs_velocity(envs_idx)

    def set_pos(self, pos, envs_idx=None, zero_velocity: bool = True):
        self._set("set_pos"); self._e.set_pos(pos, envs_idx, zero_velocity and not self._scene.sloppy_setters)

    def set_quat(self, quat, envs_idx=None, zero_velocity: bool = True):
        self._set("set_quat"); self._e.set_quat(quat, envs_idx, zero_velocity and not self._scene.sloppy_setters)


class _Collider:
    def __init__(self, scene):
        self._scene = scene

    def get_contacts(self, as_tensor: bool = True, to_torch: bool = True):
        sc, sim = self._scene, self._scene._sim
        sc.calls["get_contacts"] += 1
        force, pos, la, lb = sim.contact_force, sim.contact_pos, sim.link_a, sim.link_b
        if sc.trim_contacts and la.shape[1] > 0:
            # Genesis' collider: an env's contacts sit at the front of its row and the arrays are padded to the tick's LARGEST contact
            # count — their second dimension changes from tick to tick (contact_manager.py:391-426 sizes everything from the fresh
            # tensors).  Compaction keeps the slot order, so every per-link sum adds the same values in the same order.
            active = (la >= 0) | (lb >= 0)
            order = torch.argsort((~active).to(torch.int8), dim=1, stable=True)
            cmax = int(active.sum(1).max())
            sc.contact_counts.append(cmax)
            idx = order[:, :cmax]
            force = torch.gather(force, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
            pos = torch.gather(pos, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
            la, lb = torch.gather(la, 1, idx), torch.gather(lb, 1, idx)
        return {"force": sc._hand_out(None, force.clone()), "position": sc._hand_out(None, pos.clone()),
                "link_a": sc._hand_out(None, la.clone()), "link_b": sc._hand_out(None, lb.clone())}


class _RigidSolver:
    def __init__(self, scene):
        self._scene = scene
        self.collider = _Collider(scene)

    def get_links_quat(self):
        return self._scene._hand_out("get_links_quat", self._scene._sim.links_quat.clone())


class GenesisLikeScene:
    """``gs.Scene`` look-alike (same constructor keywords as the synthetic scene, which it drives in private)."""

    poison = True
    sloppy_setters = False  # True: set_pos / set_quat IGNORE zero_velocity — a simulator the masked reset's assumption does not hold for
    trim_contacts = False   # True: get_contacts() pads to the tick's largest contact count, as Genesis does (the shape varies per tick)

    def __init__(self, **kw):
        self.contact_counts: list = []
        self._sim = _synth.SyntheticScene(**kw)
        self.dt = self._sim.dt
        self.substeps = self._sim.substeps
        self.rigid_solver = _RigidSolver(self)
        self.calls: collections.Counter = collections.Counter()
        self._wrapped: dict = {}
        self._handed: list = [[], [], []]   # tensors handed out in this tick, the previous one, the one before
        self.is_built = False

    # -- construction -----------------------------------------------------------------------------------------------------
    def add_entity(self, morph=None, model=None, **kw):
        inner = self._sim.add_entity(morph=morph, model=model, **kw)
        if isinstance(inner, _synth.SyntheticEntity):
            ent = GenesisLikeEntity(self, inner)
            self._wrapped[id(inner)] = ent
            return ent
        return inner   # plane / terrain: static, read at build time only (TerrainManager)

    def add_camera(self, **kw):
        return self._sim.add_camera(**kw)

    def build(self, n_envs: int = 1, **kw):
        self._sim.build(n_envs=n_envs, **kw)
        self.n_envs = n_envs
        self.envs_offset = self._sim.envs_offset
        self.is_built = True

    @property
    def env_offset(self):
        return self._sim.env_offset

    @env_offset.setter
    def env_offset(self, v):
        self._sim.env_offset = v

    # -- the tick ---------------------------------------------------------------------------------------------------------
    def _hand_out(self, name, t: torch.Tensor) -> torch.Tensor:
        if name is not None:
            self.calls[name] += 1
        if self.poison:
            self._handed[0].append(t)
        return t

    def step(self):
        self.calls["step"] += 1
        b = nat.get_backend()
        tracer, b.tracer = b.tracer, None   # the simulator's own kernels are not part of the manager step being recorded
        try:
            self._sim.step()
        finally:
            b.tracer = tracer
        if self.poison:
            for t in self._handed[2]:
                if t.is_floating_point():
                    t.fill_(float("nan"))
                else:
                    t.fill_(-12345)
            self._handed = [[], self._handed[0], self._handed[1]]

    # viewer / debug API accepted and ignored
    def draw_debug_arrow(self, *a, **k):
        return None

    def draw_debug_spheres(self, *a, **k):
        return None

    def clear_debug_object(self, *a, **k):
        pass
