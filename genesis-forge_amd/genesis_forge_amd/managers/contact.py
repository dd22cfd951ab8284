"""
ContactManager — API mirror of genesis_forge/managers/contact/contact_manager.py; the whole
``step`` (NaN sanitising, the Taichi accumulation kernel of contact/kernel.py:5-90, force-norm
threshold and the four air-time ``torch.where`` updates of :434-477) is one ``gf_contact_step``
launch.  Link-name regex resolution stays Python (build time only).  Debug spheres: out of scope.
"""
from __future__ import annotations

import re
from typing import Optional

import torch

from .. import _native as nat
from .. import gs
from .base import BaseManager


class ContactManager(BaseManager):
    """Tracks contact forces on a set of links (ctor as contact_manager.py:101-156)."""

    _fused_reset = True

    def __init__(self, env, link_names: list[str], entity_attr: str = "robot", with_entity_attr: str = None,
                 with_links_names: list[str] = None, track_air_time: bool = False, air_time_contact_threshold: float = 1.0,
                 debug_visualizer: bool = False, debug_visualizer_cfg: dict | None = None):
        super().__init__(env, "contact")
        self._link_names = link_names
        self._air_time_contact_threshold = air_time_contact_threshold
        self._track_air_time = track_air_time
        self._entity_attr = entity_attr
        self._link_ids = None
        self._local_link_ids = None
        self._with_entity_attr = with_entity_attr
        self._with_links_names = with_links_names
        self._with_link_ids = torch.empty(0, device=gs.device)
        self._with_local_link_ids = None
        self._has_with_filter = with_entity_attr is not None or with_links_names is not None
        self.debug_visualizer = debug_visualizer
        self.visualizer_cfg = dict(debug_visualizer_cfg or {})
        self._contact_position_counts = None
        self.contacts: Optional[torch.Tensor] = None
        self.contact_positions: Optional[torch.Tensor] = None
        self.last_air_time: Optional[torch.Tensor] = None
        self.current_air_time: Optional[torch.Tensor] = None
        self.last_contact_time: Optional[torch.Tensor] = None
        self.current_contact_time: Optional[torch.Tensor] = None
        self._args = nat.GfContactArgs()

    @property
    def link_ids(self) -> torch.Tensor:
        return self._link_ids

    @property
    def local_link_ids(self) -> torch.Tensor:
        return self._local_link_ids

    # -- helpers (contact_manager.py:198-269) ---------------------------------------------------------
    def has_made_contact(self, dt: float, time_margin: float = 1.0e-8) -> torch.Tensor:
        if not self._track_air_time:
            raise RuntimeError("The contact sensor is not configured to track air time."
                               "Please enable the 'track_air_time' in the manager configuration.")
        return (self.current_contact_time > 0.0) * (self.current_contact_time < (dt + time_margin))

    def has_broken_contact(self, dt: float, time_margin: float = 1.0e-8) -> torch.Tensor:
        if not self._track_air_time:
            raise RuntimeError("The contact manager is not configured to track air time."
                               "Please enable the 'track_air_time' in the manager configuration.")
        return (self.current_air_time > 0.0) * (self.current_air_time < (dt + time_margin))

    def get_contact_forces(self, link_idx: int) -> torch.Tensor:
        idx = torch.nonzero(self._link_ids == link_idx)[0]
        return self.contacts[:, idx, :]

    # -- operations -----------------------------------------------------------------------------------
    def build(self):
        super().build()
        self._link_ids, self._local_link_ids = self._get_links_idx(self._entity_attr, self._link_names)
        if self._with_entity_attr or self._with_links_names:
            with_attr = self._with_entity_attr if self._with_entity_attr is not None else "robot"
            self._with_link_ids, self._with_local_link_ids = self._get_links_idx(with_attr, self._with_links_names)
        self._local_ids_list = [int(v) for v in self._local_link_ids.tolist()]   # host copy (a .tolist() per step would be a sync)
        L = self._link_ids.shape[0]
        if L > nat.GF_MAX_LINK_IDS or self._with_link_ids.shape[0] > nat.GF_MAX_LINK_IDS:
            raise RuntimeError(f"ContactManager tracks at most {nat.GF_MAX_LINK_IDS} links")
        N = self.env.num_envs
        self.contacts = torch.zeros((N, L, 3), device=gs.device)
        self.contact_positions = torch.zeros((N, L, 3), device=gs.device)
        self._contact_position_counts = torch.zeros((N, L), device=gs.device)
        self.link_vel = torch.zeros((N, L, 3), device=gs.device)  # velocity of each tracked link, refreshed by step()
        self.link_pos = torch.zeros((N, L, 3), device=gs.device)  # world position of each tracked link, refreshed by step()
        self._has_link_vel = False
        self._has_link_pos = False
        #: set by view(): some term reads this manager's link velocities / positions.  Only then does step() keep the compact
        #: per-manager copy — the gait task's 9 thigh / calf / base links do not need one (24 B read + 24 B written per link and env)
        self._want_link_vel = False
        self._want_link_pos = False
        if self._track_air_time:
            self.last_air_time = torch.zeros((N, L), device=gs.device)
            self.current_air_time = torch.zeros_like(self.last_air_time)
            self.last_contact_time = torch.zeros_like(self.last_air_time)
            self.current_contact_time = torch.zeros_like(self.last_air_time)
        a = self._args
        a.num_envs, a.num_targets, a.num_with = N, L, int(self._with_link_ids.shape[0])
        a.has_with_filter = 1 if self._has_with_filter else 0
        a.track_air_time = 1 if self._track_air_time else 0
        for i, v in enumerate(self._link_ids.tolist()):
            a.target_link_ids[i] = int(v)
        for i, v in enumerate(self._with_link_ids.tolist()):
            a.with_link_ids[i] = int(v)
        a.air_time_threshold = float(self._air_time_contact_threshold)

    def reset(self, envs_idx: list[int] | None = None):
        """contact_manager.py:316-329"""
        if not self.enabled or not self._track_air_time:
            return
        a = nat.GfResetArgs()
        a.num_envs = self.env.num_envs
        mask = self.env._ids_to_mask(envs_idx)
        a.mask = mask.data_ptr()
        self._fill_reset(a)
        self.env.backend.call("masked_reset", a)

    def _fill_reset(self, a: nat.GfResetArgs) -> None:
        if not self.enabled or not self._track_air_time:
            return
        m = a.num_contact
        if m >= nat.GF_MAX_CONTACT_VIEWS:
            raise RuntimeError(f"at most {nat.GF_MAX_CONTACT_VIEWS} air-time tracking ContactManagers")
        for s, t in enumerate((self.last_air_time, self.current_air_time, self.last_contact_time, self.current_contact_time)):
            a.air_state[m][s] = t.data_ptr()
        a.air_links[m] = self.contacts.shape[1]
        a.num_contact = m + 1

    def step(self):
        """contact_manager.py:331-336 → one launch."""
        if not self.enabled:
            return
        env = self.env
        solver = env.scene.rigid_solver
        ad = env._adapter
        if hasattr(solver, "gf_contacts"):
            c = solver.gf_contacts()  # synthetic scene: persistent buffers
        elif ad is not None:
            # Genesis-shaped scene: ONE get_contacts() / get_links_quat() per tick for all ContactManagers (the reference calls
            # both per manager, contact_manager.py:391-404), so consecutive managers also share one launch
            c = dict(ad.contacts())
            c["links_quat"] = ad.links_quat()
            for what in ("vel", "pos"):
                t = ad.solver_links(what)
                if t is not None:
                    c["links_" + what] = t
        else:
            c = solver.collider.get_contacts(as_tensor=True, to_torch=True)
        force = c["force"].to(torch.float32).contiguous()
        position = c["position"].to(torch.float32).contiguous()
        link_a = c["link_a"].to(torch.int32).contiguous()
        link_b = c["link_b"].to(torch.int32).contiguous()
        links_quat = c["links_quat"] if "links_quat" in c else solver.get_links_quat()
        links_quat = links_quat.to(torch.float32).contiguous()
        self._keep = (force, position, link_a, link_b, links_quat)
        a = self._args
        a.num_contacts = int(link_a.shape[-1]) if link_a.dim() > 1 else 0
        a.num_scene_links = int(links_quat.shape[1])
        a.force, a.position = force.data_ptr(), position.data_ptr()
        a.link_a, a.link_b, a.links_quat = link_a.data_ptr(), link_b.data_ptr(), links_quat.data_ptr()
        lv = c.get("links_vel") if isinstance(c, dict) else None
        if lv is None and ad is None and hasattr(solver, "get_links_vel"):
            lv = solver.get_links_vel()
        if lv is not None:
            lv = lv.to(torch.float32).contiguous()
            self._keep = self._keep + (lv,)
            # (the scene array is named in every manager's descriptor — managers that share the arrays share one launch — the copy
            # only where a consumer exists)
            a.links_vel, a.link_vel_out = lv.data_ptr(), (self.link_vel.data_ptr() if self._want_link_vel else None)
            self._has_link_vel = self._want_link_vel
        else:
            a.links_vel = a.link_vel_out = None
            self._has_link_vel = False
        lp = c.get("links_pos") if isinstance(c, dict) else None
        if lp is None and ad is None and hasattr(solver, "get_links_pos"):
            lp = solver.get_links_pos()
        if lp is not None:
            lp = lp.to(torch.float32).contiguous()
            self._keep = self._keep + (lp,)
            a.links_pos, a.link_pos_out = lp.data_ptr(), (self.link_pos.data_ptr() if self._want_link_pos else None)
            self._has_link_pos = self._want_link_pos
        else:
            a.links_pos = a.link_pos_out = None
            self._has_link_pos = False
        a.dt = float(env.scene.dt)
        a.contacts = self.contacts.data_ptr()
        a.contact_positions = self.contact_positions.data_ptr()
        a.position_counts = self._contact_position_counts.data_ptr()
        if self._track_air_time:
            a.last_air_time, a.current_air_time = self.last_air_time.data_ptr(), self.current_air_time.data_ptr()
            a.last_contact_time, a.current_contact_time = self.last_contact_time.data_ptr(), self.current_contact_time.data_ptr()
        a.stats = env.stats.ptr
        env.backend.call("contact_step", a, owner=self)

    def view(self, v: nat.GfContactView, need_link_vel: bool = False, need_link_pos: bool = False) -> tuple:
        """Fill a GfContactView for term kernels; returns tensors to keep alive."""
        v.contacts = self.contacts.data_ptr()
        v.num_links = self.contacts.shape[1]
        v.last_air_time = None if self.last_air_time is None else self.last_air_time.data_ptr()
        v.current_contact_time = None if self.current_contact_time is None else self.current_contact_time.data_ptr()
        keep = ()
        self._want_link_vel = self._want_link_vel or need_link_vel   # from the next step() on the launch keeps a compact copy
        self._want_link_pos = self._want_link_pos or need_link_pos
        if need_link_vel and self._has_link_vel:
            v.link_vel = self.link_vel.data_ptr()   # persistent, filled by gf_contact_step
        elif need_link_vel and self.env._adapter is not None:
            # this tick's snapshot (a recorded step patches the pointer): not a per-launch temporary, so `keep` stays empty
            v.link_vel = self.env._adapter.entity_links(getattr(self.env, self._entity_attr), "vel", self._local_ids_list).data_ptr()
        elif need_link_vel:
            robot = getattr(self.env, self._entity_attr)
            lv = robot.get_links_vel(links_idx_local=self._local_link_ids).to(torch.float32).contiguous()
            v.link_vel = lv.data_ptr()
            keep = (lv,)
        else:
            v.link_vel = None
        if need_link_pos and self._has_link_pos:
            v.link_pos = self.link_pos.data_ptr()   # persistent, filled by gf_contact_step
        elif need_link_pos and self.env._adapter is not None:
            v.link_pos = self.env._adapter.entity_links(getattr(self.env, self._entity_attr), "pos", self._local_ids_list).data_ptr()
        elif need_link_pos:
            robot = getattr(self.env, self._entity_attr)
            lp = robot.get_links_pos(links_idx_local=self._local_link_ids).to(torch.float32).contiguous()
            v.link_pos = lp.data_ptr()
            keep = keep + (lp,)
        else:
            v.link_pos = None
        return keep

    # -- implementation -----------------------------------------------------------------------------
    def _get_links_idx(self, entity_attr: str, names: list[str] = None):
        """contact_manager.py:342-382"""
        entity = self.env.__getattribute__(entity_attr)
        ids, local_ids = [], []
        if names is None:
            for link in entity.links:
                ids.append(link.idx)
                local_ids.append(link.idx_local)
        else:
            for pattern in names:
                found = False
                for link in entity.links:
                    if pattern == link.name or re.match(f"^{pattern}$", link.name):
                        ids.append(link.idx)
                        local_ids.append(link.idx_local)
                        found = True
                if not found:
                    avail = [link.name for link in entity.links]
                    raise RuntimeError(f"Link '{pattern}' not found in entity '{self._entity_attr}'.\nAvailable links: {avail}")
        return torch.tensor(ids, device=gs.device), torch.tensor(local_ids, device=gs.device)

    def __repr__(self):
        attrs = [f"link_names={self._link_names}"]
        if self._entity_attr:
            attrs.append(f"entity_attr={self._entity_attr}")
        if self._track_air_time:
            attrs.append(f"track_air_time={self._track_air_time}")
        return f"{self.__class__.__name__}({', '.join(attrs)})"


# -- annotation types of the reference (contact/config.py:4-26) ----------------------------------------------------------------------
from typing import Tuple, TypedDict  # noqa: E402


class ContactDebugVisualizerConfig(TypedDict, total=False):
    """Options of the contact markers (drawing is Genesis' viewer: out of scope here; the dict is accepted and kept)."""
    envs_idx: list
    color: Tuple[float, float, float, float]
    radius: float
    force_threshold: float


DEFAULT_VISUALIZER_CONFIG: ContactDebugVisualizerConfig = {"envs_idx": None, "size": 0.03, "color": (0.5, 0.0, 0.0, 1.0), "force_threshold": 1.0}
