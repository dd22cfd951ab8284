"""
Synthetic scene — the slice of the Genesis API the manager stack calls (SURVEY.md §8b "What it
calls"), backed by ``gf_synth_scene_step`` instead of a rigid-body solver.

It exists so the manager pipeline can be benchmarked and parity-tested where Genesis is not
installed (this image, the GPU box).  It is NOT physics: joints follow their PD targets with a
first-order lag and the base does a damped random walk (Philox driven, deterministic).  The entity
surface (``get_pos`` … ``set_dofs_position(envs_idx=…)``) follows Genesis' RigidEntity so task
configs written for the reference run unchanged; in addition the entity exposes zero-copy
``gf_*`` views that the fused phases use to avoid per-step tensor copies, and masked setters that
let the reset phase run without the ``nonzero()`` host sync real Genesis setters need.
"""
from __future__ import annotations

import ctypes as C

from dataclasses import dataclass, field
from typing import Optional

import torch

from . import _native as nat
from . import gs
from .genesis_env import EntityViews


@dataclass
class Joint:
    name: str
    type: int
    dof_start: int


@dataclass
class Link:
    name: str
    idx: int
    idx_local: int
    entity: object = field(default=None, repr=False, compare=False)

    def get_vel(self, envs_idx=None):
        """World-frame velocity of this link, ``[N,3]`` (Genesis ``RigidLink.get_vel``; call site
        examples/gait_trainer/gait_command_manager.py:329)."""
        return self.entity.get_links_vel(links_idx_local=[self.idx_local])[:, 0]

    def get_pos(self, envs_idx=None):
        return self.entity.get_links_pos(links_idx_local=[self.idx_local])[:, 0]


@dataclass
class RobotModel:
    """Kinematic description: actuated joint names (with limits) and link names."""
    name: str
    joints: list            # [(name, lower, upper)]
    links: list             # [name]
    init_pos: tuple = (0.0, 0.0, 0.4)
    init_quat: tuple = (1.0, 0.0, 0.0, 0.0)


def go2_model() -> RobotModel:
    """Unitree Go2: 12 revolute joints FL/FR/RL/RR × hip/thigh/calf (examples/simple/environment.py:122-135)."""
    joints, links = [], ["base"]
    for leg in ("FL", "FR", "RL", "RR"):
        rear = leg[0] == "R"
        joints.append((f"{leg}_hip_joint", -1.0472, 1.0472))
        joints.append((f"{leg}_thigh_joint", -0.5236 if rear else -1.5708, 4.5379 if rear else 3.4907))
        joints.append((f"{leg}_calf_joint", -2.7227, -0.83776))
        links += [f"{leg}_hip", f"{leg}_thigh", f"{leg}_calf", f"{leg}_foot"]
    return RobotModel("go2", joints, links)


def humanoid_model(num_dofs: int = 12) -> RobotModel:
    """Berkeley-humanoid-like biped.  The reference's MJCF has 12 actuated joints
    (examples/berkeley_humanoid/model/berkeley_humanoid.xml:72-133); BASELINE's "~28-DOF" variant is synthetic."""
    names12 = [f"{s}{j}" for s in ("LL_", "LR_") for j in ("HR", "HAA", "HFE", "KFE", "FFE", "FAA")]
    names = names12 if num_dofs == 12 else [f"J{k:02d}" for k in range(num_dofs)]
    joints = [(n, -1.5, 1.5) for n in names]
    links = ["torso"] + [f"{s}{l}" for s in ("ll_", "lr_") for l in ("hr", "haa", "hfe", "kfe", "ffe", "faa")]
    return RobotModel("humanoid", joints, links, init_pos=(0.0, 0.0, 0.55))


class Morph:
    def __init__(self, kind: str, **kw):
        self.kind = kind
        self.kw = kw


class morphs:
    """``gs.morphs`` look-alike: ``Plane()``, ``URDF(file=…, pos=…, quat=…)``, ``MJCF(file=…)``."""

    @staticmethod
    def Plane(**kw):
        return Morph("plane", **kw)

    @staticmethod
    def URDF(**kw):
        return Morph("urdf", **kw)

    @staticmethod
    def MJCF(**kw):
        return Morph("mjcf", **kw)

    @staticmethod
    def Terrain(**kw):
        return Morph("terrain", **kw)


class SyntheticPlane:
    """Static ground entity (one link, global index 0 when added first)."""

    def __init__(self, scene, link_start: int):
        self.scene = scene
        self.links = [Link("plane", link_start, 0)]
        self.joints = []
        self.bounds = (-50.0, 50.0, -50.0, 50.0)
        self.n_links = 1


class _TerrainMorph:
    """The attributes of ``gs.morphs.Terrain`` that TerrainManager reads (terrain_manager.py:281-359)."""

    def __init__(self, pos=(0.0, 0.0, 0.0), n_subterrains=(1, 1), subterrain_size=(12.0, 12.0), horizontal_scale=0.25,
                 vertical_scale=0.005, subterrain_types="flat_terrain", subterrain_parameters=None, height_field=None, **_ignored):
        self.pos = tuple(float(v) for v in pos)
        self.n_subterrains = tuple(int(v) for v in n_subterrains)
        self.subterrain_size = tuple(float(v) for v in subterrain_size)
        self.horizontal_scale = float(horizontal_scale)
        self.vertical_scale = float(vertical_scale)
        if isinstance(subterrain_types, str):
            subterrain_types = [[subterrain_types] * self.n_subterrains[1] for _ in range(self.n_subterrains[0])]
        self.subterrain_types = subterrain_types
        self.subterrain_parameters = subterrain_parameters or {}
        self.height_field = height_field


class _TerrainGeom:
    def __init__(self, metadata, lo, hi, pos):
        self.metadata = metadata
        self._aabb = torch.tensor([lo, hi], dtype=torch.float32)
        self._pos = torch.tensor(pos, dtype=torch.float32)

    def get_AABB(self):
        return self._aabb

    def get_pos(self):
        return self._pos


class SyntheticTerrain(SyntheticPlane):
    """``scene.add_entity(morph=gs.morphs.Terrain(…))``: a static height-field entity.

    The height field has Genesis' layout — integer height steps, shape ``(n_sub_x * rows, n_sub_y * cols)`` with
    ``rows = subterrain_size[0] / horizontal_scale``, metres = value * vertical_scale — and is filled per subterrain type:
    ``flat_terrain`` zeros; anything else a deterministic blocky random field between the type's ``min_height`` and
    ``max_height`` (a stand-in for Genesis' generators: TerrainManager only ever samples the field, it never generates it)."""

    def __init__(self, scene, link_start: int, morph: "Morph"):
        super().__init__(scene, link_start)
        import numpy as np

        m = _TerrainMorph(**morph.kw)
        self.morph = m
        rows = int(m.subterrain_size[0] / m.horizontal_scale + 1e-9)
        cols = int(m.subterrain_size[1] / m.horizontal_scale + 1e-9)
        if m.height_field is not None:
            hf = np.asarray(m.height_field, dtype=np.int16)
        else:
            hf = np.zeros((m.n_subterrains[0] * rows, m.n_subterrains[1] * cols), dtype=np.int16)
            rng = np.random.RandomState(scene.seed & 0x7FFFFFFF)
            for i in range(m.n_subterrains[0]):
                for j in range(m.n_subterrains[1]):
                    kind = m.subterrain_types[i][j]
                    if kind == "flat_terrain":
                        continue
                    par = m.subterrain_parameters.get(kind, {})
                    lo = int(round(par.get("min_height", 0.0) / m.vertical_scale))
                    hi = max(lo, int(round(par.get("max_height", 0.1) / m.vertical_scale)))
                    blk = 4
                    coarse = rng.randint(lo, hi + 1, size=((rows + blk - 1) // blk, (cols + blk - 1) // blk))
                    hf[i * rows:(i + 1) * rows, j * cols:(j + 1) * cols] = np.kron(coarse, np.ones((blk, blk), dtype=np.int64))[:rows, :cols]
        x0, y0, z0 = m.pos
        lo = (x0, y0, z0 + float(hf.min()) * m.vertical_scale)
        hi = (x0 + m.n_subterrains[0] * m.subterrain_size[0], y0 + m.n_subterrains[1] * m.subterrain_size[1], z0 + float(hf.max()) * m.vertical_scale)
        self.geoms = [_TerrainGeom({"height_field": hf}, lo, hi, m.pos)]
        self.bounds = (lo[0], hi[0], lo[1], hi[1])
        self.links = [Link("terrain", link_start, 0)]


class SyntheticEntity:
    """Articulated robot with a free base; state lives in ``[N, …]`` device tensors."""

    def __init__(self, scene, model: RobotModel, link_start: int, pos=None, quat=None):
        self.scene = scene
        self.model = model
        self.init_pos = tuple(pos) if pos is not None else model.init_pos
        self.init_quat = tuple(quat) if quat is not None else model.init_quat
        self.joints = [Joint("root_joint", int(gs.JOINT_TYPE.FREE), 0)]
        for k, (name, lo, hi) in enumerate(model.joints):
            self.joints.append(Joint(name, int(gs.JOINT_TYPE.REVOLUTE), 6 + k))
        self.links = [Link(n, link_start + k, k, self) for k, n in enumerate(model.links)]
        self.n_links = len(self.links)
        self.n_act = len(model.joints)
        self.n_dofs = 6 + self.n_act
        self._lower = torch.tensor([j[1] for j in model.joints], dtype=torch.float32)
        self._upper = torch.tensor([j[2] for j in model.joints], dtype=torch.float32)
        self._built = False
        self._cols_cache: dict = {}
        self.gains: dict = {}

    # -- allocation ---------------------------------------------------------------------------------
    def _build(self, n: int):
        dev = gs.device
        self.n_envs = n
        self.pos = torch.tensor(self.init_pos, device=dev, dtype=torch.float32).repeat(n, 1).contiguous()
        self.quat = torch.tensor(self.init_quat, device=dev, dtype=torch.float32).repeat(n, 1).contiguous()
        self.lin_vel = torch.zeros(n, 3, device=dev)
        self.ang_vel = torch.zeros(n, 3, device=dev)
        self.dof_pos = torch.zeros(n, self.n_act, device=dev)
        self.dof_vel = torch.zeros(n, self.n_act, device=dev)
        self.dof_force = torch.zeros(n, self.n_act, device=dev)
        self._own_targets = torch.zeros(n, self.n_act, device=dev)
        self._targets = self._own_targets
        self.links_vel = torch.zeros(n, self.n_links, 3, device=dev)
        self.links_pos = None  # view of the scene's per-link positions once the scene tick produces them
        self._lower = self._lower.to(dev)
        self._upper = self._upper.to(dev)
        self._views = EntityViews(self.pos, self.quat, self.lin_vel, self.ang_vel)
        self._built = True

    # -- zero-copy fast paths used by the fused phases -------------------------------------------------
    def gf_views(self) -> EntityViews:
        return self._views

    def _cols(self, dofs_idx) -> Optional[list[int]]:
        if dofs_idx is None:
            return None
        key = tuple(dofs_idx.tolist() if isinstance(dofs_idx, torch.Tensor) else dofs_idx)
        hit = self._cols_cache.get(key)
        if hit is None:
            cols = [int(i) - 6 for i in key]
            hit = self._cols_cache[key] = (None if cols == list(range(self.n_act)) else cols,)
        return hit[0]

    def gf_dofs(self, what: str, dofs_idx) -> torch.Tensor:
        t = {"position": self.dof_pos, "velocity": self.dof_vel, "force": self.dof_force}[what]
        cols = self._cols(dofs_idx)
        return t if cols is None else t[:, cols].contiguous()

    def gf_masked_dofs(self, dofs_idx):
        if self._cols(dofs_idx) is not None:
            raise RuntimeError("masked DOF reset needs the action manager to control every actuated joint in order")
        return self.dof_pos, self.dof_vel

    def gf_masked_base(self):
        return self.pos, self.quat, self.lin_vel, self.ang_vel

    # -- Genesis RigidEntity getters (fresh tensors, like Genesis) --------------------------------------
    def get_pos(self, envs_idx=None):
        return self.pos.clone() if envs_idx is None else self.pos[envs_idx]

    def get_quat(self, envs_idx=None):
        return self.quat.clone() if envs_idx is None else self.quat[envs_idx]

    def get_vel(self, envs_idx=None):
        return self.lin_vel.clone() if envs_idx is None else self.lin_vel[envs_idx]

    def get_ang(self, envs_idx=None):
        return self.ang_vel.clone() if envs_idx is None else self.ang_vel[envs_idx]

    def get_dofs_position(self, dofs_idx_local=None, envs_idx=None):
        cols = self._cols(dofs_idx_local)
        return self.dof_pos.clone() if cols is None else self.dof_pos[:, cols]

    def get_dofs_velocity(self, dofs_idx_local=None, envs_idx=None):
        cols = self._cols(dofs_idx_local)
        return self.dof_vel.clone() if cols is None else self.dof_vel[:, cols]

    def get_dofs_force(self, dofs_idx_local=None, envs_idx=None):
        cols = self._cols(dofs_idx_local)
        return self.dof_force.clone() if cols is None else self.dof_force[:, cols]

    def get_dofs_limit(self, dofs_idx_local=None):
        cols = self._cols(dofs_idx_local)
        if cols is None:
            return self._lower.clone(), self._upper.clone()
        return self._lower[cols], self._upper[cols]

    def get_links_vel(self, links_idx_local=None, envs_idx=None):
        if links_idx_local is None:
            return self.links_vel.clone()
        idx = links_idx_local.tolist() if isinstance(links_idx_local, torch.Tensor) else list(links_idx_local)
        return self.links_vel[:, idx]

    def get_links_pos(self, links_idx_local=None, envs_idx=None):
        if self.links_pos is None:
            n = self.n_links if links_idx_local is None else len(links_idx_local)
            return self.pos.unsqueeze(1).expand(-1, n, -1).clone()
        if links_idx_local is None:
            return self.links_pos.clone()
        idx = links_idx_local.tolist() if isinstance(links_idx_local, torch.Tensor) else list(links_idx_local)
        return self.links_pos[:, idx]

    def get_link(self, name: str):
        for l in self.links:
            if l.name == name:
                return l
        raise KeyError(name)

    def get_AABB(self):
        lo = self.pos - 0.3
        hi = self.pos + 0.3
        return torch.stack([lo, hi], dim=1)

    # -- control & setters ---------------------------------------------------------------------------
    def control_dofs_position(self, position, dofs_idx_local=None, envs_idx=None):
        cols = self._cols(dofs_idx_local)
        if cols is None and envs_idx is None and position.shape == self._own_targets.shape and position.dtype == torch.float32 \
                and position.is_contiguous():
            self._targets = position  # zero copy: the scene tick reads the action manager's target buffer
        elif cols is None:
            self._own_targets[:] = position
            self._targets = self._own_targets
        else:
            self._own_targets[:, cols] = position
            self._targets = self._own_targets

    def _store_gain(self, name, value, dofs_idx_local, envs_idx):
        self.gains[name] = value

    def set_dofs_kp(self, kp, dofs_idx_local=None, envs_idx=None):
        self._store_gain("kp", kp, dofs_idx_local, envs_idx)

    def set_dofs_kv(self, kv, dofs_idx_local=None, envs_idx=None):
        self._store_gain("kv", kv, dofs_idx_local, envs_idx)

    def set_dofs_damping(self, v, dofs_idx_local=None, envs_idx=None):
        self._store_gain("damping", v, dofs_idx_local, envs_idx)

    def set_dofs_stiffness(self, v, dofs_idx_local=None, envs_idx=None):
        self._store_gain("stiffness", v, dofs_idx_local, envs_idx)

    def set_dofs_frictionloss(self, v, dofs_idx_local=None, envs_idx=None):
        self._store_gain("frictionloss", v, dofs_idx_local, envs_idx)

    def set_dofs_force_range(self, lower, upper, dofs_idx_local=None, envs_idx=None):
        self._store_gain("force_range", (lower, upper), dofs_idx_local, envs_idx)

    def set_mass_shift(self, shift, links_idx_local=None, envs_idx=None):
        self._store_gain("mass_shift", shift, links_idx_local, envs_idx)

    @staticmethod
    def _rows(envs_idx):
        return slice(None) if envs_idx is None else envs_idx

    def set_dofs_position(self, position, dofs_idx_local=None, envs_idx=None, zero_velocity: bool = True):
        rows, cols = self._rows(envs_idx), self._cols(dofs_idx_local)
        if cols is None:
            self.dof_pos[rows] = position
            if zero_velocity:
                self.dof_vel[rows] = 0.0
        else:
            sub = self.dof_pos[rows]
            sub[:, cols] = position
            self.dof_pos[rows] = sub

    def zero_all_dofs_velocity(self, envs_idx=None):
        rows = self._rows(envs_idx)
        self.dof_vel[rows] = 0.0
        self.lin_vel[rows] = 0.0
        self.ang_vel[rows] = 0.0

    def set_pos(self, pos, envs_idx=None, zero_velocity: bool = True):
        self.pos[self._rows(envs_idx)] = pos
        if zero_velocity:
            self.zero_all_dofs_velocity(envs_idx)

    def set_quat(self, quat, envs_idx=None, zero_velocity: bool = True):
        self.quat[self._rows(envs_idx)] = quat
        if zero_velocity:
            self.zero_all_dofs_velocity(envs_idx)


class _Collider:
    def __init__(self, scene):
        self._scene = scene

    def get_contacts(self, as_tensor: bool = True, to_torch: bool = True):
        s = self._scene
        return {"force": s.contact_force.clone(), "position": s.contact_pos.clone(), "link_a": s.link_a.clone(), "link_b": s.link_b.clone()}


class SyntheticScene:
    """``gs.Scene`` look-alike driving ``gf_synth_scene_step``.  Extra keyword options are accepted and ignored
    so reference-style constructor calls (``sim_options=…, viewer_options=…``) keep working."""

    def __init__(self, dt: float = 0.02, substeps: int = 2, max_collision_pairs: int = 0, seed: int = 1234,
                 joint_rate: float = 10.0, ang_noise: float = 0.05, lin_noise: float = 0.05, height_target: Optional[float] = None,
                 contact_prob: float = 0.15, contact_force: float = 40.0, foot_links=None, foot_contact_prob: float = 0.5, **_ignored):
        self.dt = dt
        self.substeps = substeps
        self.n_contacts = int(max_collision_pairs)
        self.seed = seed
        self.joint_rate, self.ang_noise, self.lin_noise = joint_rate, ang_noise, lin_noise
        self.height_target = height_target
        self.contact_prob, self.contact_force_scale = contact_prob, contact_force
        # walking contact model (GfSynthSceneArgs.foot_link_mask): `foot_links` = regular expressions over the robot's link names;
        # the matching links touch the ground in a trot pattern, `contact_prob` is then the density of the other (body) contacts
        self.foot_links = [foot_links] if isinstance(foot_links, str) else list(foot_links or [])
        self.foot_contact_prob = foot_contact_prob
        self._foot_mask = 0
        self.entities: list = []
        self.robot: Optional[SyntheticEntity] = None
        self._n_links = 0
        self._tick_c = C.c_uint64(0)   # a ctypes cell: a recorded step's native patch table advances it (GF_PATCH_COUNTER)
        self.rigid_solver = self
        self.collider = _Collider(self)
        self._args = nat.GfSynthSceneArgs()
        self.is_built = False
        self.env_offset = 0  # global index of local env 0 (env sharding)
        self.gf_static_buffers = True  # every state tensor is allocated once at build(): a step can be recorded

    # -- construction ---------------------------------------------------------------------------------
    def add_entity(self, morph=None, model: Optional[RobotModel] = None, **kw):
        if isinstance(morph, RobotModel):
            model, morph = morph, None
        if model is None and morph is not None and morph.kind == "terrain":
            ent = SyntheticTerrain(self, self._n_links, morph)
        elif model is None and morph is not None and morph.kind == "plane":
            ent = SyntheticPlane(self, self._n_links)
        else:
            if model is None:
                f = str(morph.kw.get("file", "")) if morph is not None else ""
                model = humanoid_model() if "humanoid" in f else go2_model()
            pos = morph.kw.get("pos") if morph is not None else None
            quat = morph.kw.get("quat") if morph is not None else None
            ent = SyntheticEntity(self, model, self._n_links, pos=pos, quat=quat)
            if self.robot is None:
                self.robot = ent
        self._n_links += ent.n_links
        self.entities.append(ent)
        return ent

    def add_camera(self, **kw):
        class _Cam:
            def follow_entity(self, *a, **k):
                pass
        return _Cam()

    def build(self, n_envs: int = 1, **_ignored):
        dev = gs.device
        self.n_envs = n_envs
        for e in self.entities:
            if isinstance(e, SyntheticEntity):
                e._build(n_envs)
        C = self.n_contacts
        self.contact_force = torch.zeros(n_envs, C, 3, device=dev)
        self.contact_pos = torch.zeros(n_envs, C, 3, device=dev)
        self.link_a = torch.full((n_envs, C), -1, device=dev, dtype=torch.int32)
        self.link_b = torch.full((n_envs, C), -1, device=dev, dtype=torch.int32)
        self.links_quat = torch.zeros(n_envs, max(self._n_links, 1), 4, device=dev)
        self.links_quat[..., 0] = 1.0
        self.links_vel_all = torch.zeros(n_envs, max(self._n_links, 1), 3, device=dev)
        self.links_pos_all = torch.zeros(n_envs, max(self._n_links, 1), 3, device=dev)
        import numpy as _np
        self.envs_offset = _np.zeros((n_envs, 3), dtype=_np.float32)  # gs.Scene.envs_offset (viewer placement; velocity_command.py:244)
        if self.foot_links and self.robot is not None:
            import re
            for l in self.robot.links:
                if any(re.fullmatch(pat, l.name) for pat in self.foot_links):
                    if l.idx >= 32:
                        raise ValueError("foot_links: scene link index beyond the 32-bit foot mask")
                    self._foot_mask |= 1 << l.idx
            if not self._foot_mask:
                raise ValueError(f"foot_links {self.foot_links} match no link of the robot")
            if bin(self._foot_mask).count("1") > self.n_contacts:
                raise ValueError("foot_links: more feet than contact slots (max_collision_pairs)")
        self.is_built = True

    # -- solver surface -------------------------------------------------------------------------------
    def get_links_quat(self):
        return self.links_quat.clone()

    def gf_contacts(self) -> dict:
        return {"force": self.contact_force, "position": self.contact_pos, "link_a": self.link_a, "link_b": self.link_b,
                "links_quat": self.links_quat, "links_vel": self.links_vel_all, "links_pos": self.links_pos_all}

    def step(self):
        """One synthetic tick (stands in for managed_env.py:292)."""
        r = self.robot
        a = self._args
        a.num_envs, a.num_dofs = self.n_envs, r.n_act
        a.num_contacts, a.num_scene_links = self.n_contacts, self._n_links
        a.dt, a.joint_rate, a.ang_noise, a.lin_noise = self.dt, self.joint_rate, self.ang_noise, self.lin_noise
        a.height_target = r.init_pos[2] if self.height_target is None else self.height_target
        a.contact_prob, a.contact_force = self.contact_prob, self.contact_force_scale
        a.foot_contact_prob, a.foot_link_mask = self.foot_contact_prob, self._foot_mask
        a.targets = r._targets.data_ptr()
        a.pos, a.quat, a.lin_vel, a.ang_vel = r.pos.data_ptr(), r.quat.data_ptr(), r.lin_vel.data_ptr(), r.ang_vel.data_ptr()
        a.dof_pos, a.dof_vel = r.dof_pos.data_ptr(), r.dof_vel.data_ptr()
        if self.n_contacts > 0:
            a.contact_force_out, a.contact_pos_out = self.contact_force.data_ptr(), self.contact_pos.data_ptr()
            a.link_a_out, a.link_b_out = self.link_a.data_ptr(), self.link_b.data_ptr()
            a.links_quat_out = self.links_quat.data_ptr()
            a.links_vel_out = self.links_vel_all.data_ptr()
            a.links_pos_out = self.links_pos_all.data_ptr()
        a.seed, a.tick, a.env_offset = self.seed, self.tick, self.env_offset
        nat.get_backend().call("synth_scene_step", a, owner=self)
        if self.n_contacts > 0:
            s = r.links[0].idx
            r.links_vel = self.links_vel_all[:, s:s + r.n_links]
            r.links_pos = self.links_pos_all[:, s:s + r.n_links]
        self.tick += 1

    @property
    def tick(self) -> int:
        return self._tick_c.value

    @tick.setter
    def tick(self, v: int) -> None:
        self._tick_c.value = v

    # viewer / debug API accepted and ignored
    def draw_debug_arrow(self, *a, **k):
        return None

    def draw_debug_spheres(self, *a, **k):
        return None

    def clear_debug_object(self, *a, **k):
        pass
