"""
Scene adapter — the L0 side of the boundary for a scene that offers ONLY Genesis' public surface (SURVEY.md §8b "What it
calls"): every getter returns a fresh tensor, state is written back through ``envs_idx`` setters, contacts come from
``rigid_solver.collider.get_contacts(as_tensor=True, to_torch=True)``.  The reference talks to such a scene from every term
(``utils.py:23-24,37-38,51-55``, ``entity_manager.py:189-195``, ``position_action_manager.py:243-289``,
``contact_manager.py:384-432``), a dozen getter calls per step; here the scene is read ONCE per tick:

* **Snapshot.**  After ``scene.step()`` each piece of state a phase needs is fetched through its public getter exactly once
  (``get``), as a contiguous f32 / i32 tensor, and every manager of the tick reads that tensor.  Keys are stable across
  ticks (which entity, which getter, which index list), so the fetch PLAN of one tick — the ordered list of keys and their
  fetchers — describes every later tick.
* **Recorded step.**  A recording freezes descriptors, and a snapshot tensor is new every tick.  ``attribute()`` finds every
  pointer field of the recorded descriptors that lies inside a snapshot tensor and turns it into a
  ``GF_PATCH_PARAM_OFFSET`` entry of a patch table: a replayed step calls ``control_dofs_position`` + ``scene.step()``,
  re-fetches the plan (``refetch``) and hands the new addresses to ``gf_replay_step`` as call parameters.  Nothing is staged
  or copied: the kernels read Genesis' tensors in place.
* **Masked reset on the snapshot, setters afterwards.**  The reset phase writes the post-reset state of the done envs (default
  joint positions + noise, the spawn pose, zero velocities) into the snapshot tensors — so the observation of the same tick
  sees what Genesis' getters return after the reference's ``reset(ids)`` (``managed_env.py:322-326``) — and the simulator itself
  is brought up to date through its ``envs_idx`` setters (``position_action_manager.py:455-464``, ``mdp/reset.py:102-124``)
  with the rows of the done envs taken from the snapshot: ``push``.  That needs the index list — the ONE ``nonzero()`` of the
  step, the same sync the reference pays at ``managed_env.py:308-310`` — and in a recorded step it runs behind the last launch.
  Assumption (true of the test double, documented for Genesis): after ``set_pos / set_quat / set_dofs_position`` the getters
  return what was set and zero velocities where ``zero_velocity`` asked for it; contacts and link states keep their values
  until the next ``scene.step()``.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import torch

from . import _native as nat


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t if t.is_contiguous() else t.contiguous()


def _i32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.int32:
        t = t.to(torch.int32)
    return t if t.is_contiguous() else t.contiguous()


def pointer_fields(struct_type, base: int = 0, prefix: str = "", out: Optional[list] = None) -> list:
    """``(byte offset, dotted name)`` of every pointer-typed field (``c_void_p``) of a ctypes structure — nested structures and
    arrays included."""
    out = [] if out is None else out
    for name, ctype, *_ in struct_type._fields_:
        _walk(ctype, base + getattr(struct_type, name).offset, prefix + name, out)
    return out


def _walk(ctype, off: int, name: str, out: list) -> None:
    if ctype is C.c_void_p:
        out.append((off, name))
    elif isinstance(ctype, type) and issubclass(ctype, C.Structure):
        pointer_fields(ctype, off, name + ".", out)
    elif isinstance(ctype, type) and issubclass(ctype, C.Array):
        el = ctype._type_
        if el is C.c_void_p or (isinstance(el, type) and issubclass(el, (C.Structure, C.Array))):
            size = C.sizeof(el)
            for i in range(ctype._length_):
                _walk(el, off + i * size, f"{name}[{i}]", out)


_PTR_CACHE: dict = {}


def struct_pointer_fields(struct) -> list:
    t = type(struct)
    offs = _PTR_CACHE.get(t)
    if offs is None:
        offs = _PTR_CACHE[t] = pointer_fields(t)
    return offs


#: pointer fields a step may legitimately change without the scene being involved: the policy's action tensor, statistics ring
#: slots, output tensors / slots, parity-mode draws, Python-evaluated columns, the statistics pack / copy descriptors
PER_STEP_FIELDS = {"actions_in", "stats", "stats_zero", "stats_fold_src", "stats_fold_dst", "stats_last_reset", "obs", "prev_obs",
                   "out", "out2", "noise_draws", "draws", "len_draws", "dof_draws", "spawn_draws", "obs_out", "reward_out", "done_out",
                   "src", "dst"}


class SceneAdapter:
    """Per-env snapshot of a Genesis-shaped scene (module docstring)."""

    def __init__(self, env):
        self.env = env
        self.epoch = 0
        self._cache: dict = {}      # key -> tensor of the current epoch
        self._fetch: dict = {}      # key -> fetcher (stable across epochs)
        self.order: list = []       # keys in first-fetch order of the current epoch
        self._pushes: dict = {}     # name -> callable(ids) registered by the reset sections of this tick's reset descriptor
        self._hold = None           # the previous epoch's tensors: kept until the next one so no launch outlives its inputs
        self.fetches = 0            # getter calls made (tests count them)
        self.fetches_last_tick = 0  # getter calls of the last replayed tick (== len(plan): one call per getter and tick)
        #: None = not checked yet; True = the simulator's envs_idx setters behave as the masked reset assumes (verify_setters);
        #: False = they do not: resets go by index list through the managers' reference-style reset(ids), with a warning
        self.setters_verified: Optional[bool] = None
        self.setter_report: list = []
        self._push_checked = False

    # -- snapshot ---------------------------------------------------------------------------------------------------------
    def invalidate(self) -> None:
        """The scene may have changed (``scene.step()``, a setter called from outside): the next ``get`` fetches again."""
        self.epoch += 1
        self._hold = self._cache
        self._cache = {}
        self.order = []

    def get(self, key, fetch: Callable[[], torch.Tensor]) -> torch.Tensor:
        t = self._cache.get(key)
        if t is None:
            t = fetch()
            self.fetches += 1
            self._cache[key] = t
            self._fetch[key] = fetch
            self.order.append(key)
        return t

    def peek(self, key) -> Optional[torch.Tensor]:
        return self._cache.get(key)

    def plan(self) -> list:
        """(key, fetcher) of everything this tick's phases read, in fetch order."""
        return [(k, self._fetch[k]) for k in self.order]

    def refetch(self, plan: list) -> list:
        """A replayed tick: fetch the recorded plan afresh (after ``scene.step()``); returns the tensors in plan order."""
        self.epoch += 1
        self._hold = self._cache
        cache, order, out = {}, [], []
        for key, fetch in plan:
            t = fetch()
            cache[key] = t
            order.append(key)
            out.append(t)
        self.fetches += len(plan)
        self.fetches_last_tick = len(plan)
        self._cache, self.order = cache, order
        return out

    # -- the documented assumption, checked against the simulator at hand ------------------------------------------------------------
    def verify_setters(self, entity, dofs_idx=None, env_id: int = 0, tol: float = 1e-6) -> bool:
        """The masked reset on an adapter scene writes the done envs' post-reset rows into the tick's snapshot and brings the
        simulator up to date through its ``envs_idx`` setters (``push``); the observation of the same tick then reads the SNAPSHOT.
        That is only right if, after ``set_pos / set_quat / set_dofs_position``, the simulator's getters return what was set — and
        zero velocities where ``zero_velocity`` asked for it (mdp/reset.py:102-124, position_action_manager.py:455-464,
        entity_manager.py:189-195).  Checked once, on one env, right before the first full reset (which overwrites the probe): set a
        pose / joint row through the setters, read it back through the getters, compare, restore.  Returns False (and fills
        ``setter_report``) when the simulator behaves differently; the env then resets by index list the reference's way."""
        ids = torch.tensor([int(env_id)], device=entity.get_pos().device, dtype=torch.long)
        report = []

        def close(name, got, want):
            got, want = got.detach().float().cpu().reshape(-1), want.detach().float().cpu().reshape(-1)
            if got.shape != want.shape or not torch.allclose(got, want, atol=tol, rtol=0):
                report.append(f"{name}: read back {got.tolist()} after setting {want.tolist()}")

        try:
            pos0, quat0 = entity.get_pos().clone(), entity.get_quat().clone()
            probe_p = pos0[ids] + torch.tensor([[0.125, -0.25, 0.0625]], device=pos0.device, dtype=pos0.dtype)
            entity.set_pos(probe_p, envs_idx=ids, zero_velocity=True)
            close("set_pos -> get_pos", entity.get_pos()[ids], probe_p)
            close("set_pos(zero_velocity=True) -> get_vel", entity.get_vel()[ids], torch.zeros(1, 3))
            close("set_pos(zero_velocity=True) -> get_ang", entity.get_ang()[ids], torch.zeros(1, 3))
            probe_q = torch.tensor([[0.8, 0.0, 0.6, 0.0]], device=quat0.device, dtype=quat0.dtype)
            entity.set_quat(probe_q, envs_idx=ids, zero_velocity=True)
            close("set_quat -> get_quat", entity.get_quat()[ids], probe_q)
            close("set_quat leaves get_pos alone", entity.get_pos()[ids], probe_p)
            if dofs_idx is not None:
                idx = [int(i) for i in dofs_idx]
                d0 = entity.get_dofs_position(idx).clone()
                probe_d = d0[ids] + 0.03125
                entity.set_dofs_position(position=probe_d, dofs_idx_local=idx, envs_idx=ids)
                close("set_dofs_position -> get_dofs_position", entity.get_dofs_position(idx)[ids], probe_d)
                close("set_dofs_position -> get_dofs_velocity", entity.get_dofs_velocity(idx)[ids], torch.zeros(1, len(idx)))
                entity.set_dofs_position(position=d0[ids], dofs_idx_local=idx, envs_idx=ids)
            entity.set_pos(pos0[ids], envs_idx=ids, zero_velocity=True)
            entity.set_quat(quat0[ids], envs_idx=ids, zero_velocity=True)
        except Exception as e:   # a simulator without one of the calls: nothing to rely on
            report.append(f"{type(e).__name__}: {e}")
        self.setter_report = report
        self.setters_verified = not report
        self.invalidate()
        return self.setters_verified

    # -- typed fetchers ----------------------------------------------------------------------------------------------------
    def base(self, entity, what: str) -> torch.Tensor:
        """World-frame base state of ``entity``: ``what`` in pos / quat / vel / ang (Genesis ``RigidEntity.get_*``)."""
        return self.get(("base", id(entity), what), lambda: _f32c(getattr(entity, "get_" + what)()))

    def dofs(self, entity, what: str, dofs_idx) -> torch.Tensor:
        """``[N, D]`` state of the DOFs ``dofs_idx``: ``what`` in position / velocity / force."""
        idx = tuple(int(i) for i in dofs_idx)
        return self.get(("dofs", id(entity), what, idx), lambda: _f32c(getattr(entity, "get_dofs_" + what)(list(idx))))

    def contacts(self) -> dict:
        """The collider's contact arrays, fetched once for every ContactManager of the scene (contact_manager.py:384-392)."""
        solver = self.env.scene.rigid_solver
        state = {"raw": None, "left": 0}   # one get_contacts() call serves the four arrays of a tick

        def part(name, conv):
            def fetch():
                if state["left"] == 0:
                    state["raw"], state["left"] = solver.collider.get_contacts(as_tensor=True, to_torch=True), 4
                t = conv(state["raw"][name])
                state["left"] -= 1
                if state["left"] == 0:
                    state["raw"] = None
                return t
            return fetch

        out = {}
        for name, conv in (("force", _f32c), ("position", _f32c), ("link_a", _i32c), ("link_b", _i32c)):
            out[name] = self.get(("contacts", name), part(name, conv))
        return out

    def links_quat(self) -> torch.Tensor:
        solver = self.env.scene.rigid_solver
        return self.get(("links_quat",), lambda: _f32c(solver.get_links_quat()))

    def solver_links(self, what: str) -> Optional[torch.Tensor]:
        """Per-link world velocity / position of the whole scene when the solver offers it (``get_links_vel`` / ``get_links_pos``)."""
        solver = self.env.scene.rigid_solver
        fn = getattr(solver, "get_links_" + what, None)
        if fn is None:
            return None
        return self.get(("links_all", what), lambda: _f32c(fn()))

    def entity_links(self, entity, what: str, links_idx_local) -> torch.Tensor:
        """``[N, L, 3]`` world velocity / position of an entity's links (``RigidEntity.get_links_vel / get_links_pos``)."""
        idx = tuple(int(i) for i in links_idx_local)
        return self.get(("links", id(entity), what, idx), lambda: _f32c(getattr(entity, "get_links_" + what)(links_idx_local=list(idx))))

    # -- write-back through the envs_idx setters ------------------------------------------------------------------------------
    def begin_reset(self) -> None:
        self._pushes = {}

    def on_push(self, name: str, fn: Callable[[torch.Tensor], None]) -> None:
        """Registered by a manager's reset section: ``fn(ids)`` writes the snapshot rows of ``ids`` into the simulator."""
        self._pushes[name] = fn

    def push(self, ids: torch.Tensor) -> None:
        for fn in self._pushes.values():
            fn(ids)

    def push_done(self, mask: torch.Tensor, mask2: Optional[torch.Tensor]) -> None:
        """The in-step reset: index list of the done envs — the one synchronisation of the step, where the reference has its
        ``nonzero()`` (managed_env.py:308-310); here ``gf_done_compact`` + a stream sync — then the setters."""
        if not self._pushes:
            return
        ids = self.env.done_ids(mask, mask2)   # (a view of the env's index buffer: the setters gather with it right away)
        if ids.numel() > 0:
            self.push(ids)
            if not self._push_checked:
                self._push_checked = True
                self.verify_push(ids)

    def verify_push(self, ids: torch.Tensor, tol: float = 1e-6) -> bool:
        """The second half of the setter check, on the first in-step reset (the probe before the first full reset sees a scene at
        rest, where "velocities are zeroed" cannot fail): the rows the masked reset wrote into the snapshot for the done envs must be
        what the simulator's getters return now that the setters have run.  A mismatch switches the env to index-list resets."""
        import warnings

        report = []
        ids = ids.clone()
        for key, t in list(self._cache.items()):
            if key[0] not in ("base", "dofs"):
                continue
            fresh = self._fetch[key]()
            self.fetches += 1
            a, b = fresh[ids].detach().float().cpu(), t[ids].detach().float().cpu()
            if a.shape != b.shape or not torch.allclose(a, b, atol=tol, rtol=0):
                report.append(f"{key[0]} {key[2]}: the simulator holds values the masked reset did not write (max |d| = {float((a - b).abs().max()):.3g})")
        if report:
            self.setter_report = report
            self.setters_verified = False
            warnings.warn("genesis_forge_amd: after the envs_idx setters the simulator does not hold the reset rows the masked reset assumed ("
                          + "; ".join(report) + ") - scene-side resets fall back to the index-list path", RuntimeWarning)
            self.env._partition_cache = None
            self.env.invalidate_trace()
        return not report

    # -- pointer attribution for a recorded step ---------------------------------------------------------------------------
    def attribute(self, descriptors: list, plan: list, skip: set) -> tuple:
        """Patch entries for every pointer field of ``descriptors`` that addresses a snapshot tensor of the current epoch.

        Returns ``(patches, covered)``: ``patches`` = ``[(field address, plan index, byte offset)]``, ``covered`` = the set of
        field addresses.  ``skip``: field addresses other patches of the recording already write."""
        spans = []
        for i, (key, _f) in enumerate(plan):
            t = self._cache[key]
            lo = t.data_ptr()
            spans.append((lo, lo + max(t.numel() * t.element_size(), 1), i))
        spans.sort()
        patches, covered = [], set()
        for d in descriptors:
            base = C.addressof(d)
            raw = C.string_at(base, C.sizeof(d))
            for off, _name in struct_pointer_fields(d):
                addr = base + off
                if addr in skip:
                    continue
                v = int.from_bytes(raw[off:off + 8], "little")
                if not v:
                    continue
                for lo, hi, i in spans:
                    if lo <= v < hi:
                        patches.append((addr, i, v - lo))
                        covered.add(addr)
                        break
        return patches, covered


def changed_pointer_fields(descriptors: list, before: list, now: list) -> list:
    """``(field address, "Struct.field")`` of the pointer fields whose value at launch time differs between two consecutive
    steps (``before[i]`` / ``now[i]``: byte images of ``descriptors[i]`` taken when it was launched)."""
    out = []
    for d, old, new in zip(descriptors, before, now):
        if old is None or new is None or len(old) != len(new):
            continue
        base = C.addressof(d)
        for off, name in struct_pointer_fields(d):
            if new[off:off + 8] != old[off:off + 8]:
                out.append((base + off, f"{type(d).__name__}.{name}"))
    return out
