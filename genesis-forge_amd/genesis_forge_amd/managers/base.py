"""BaseManager — registration + lifecycle (mirror of genesis_forge/managers/base.py:4-43)."""
from __future__ import annotations

from typing import Literal

ManagerType = Literal["action", "reward", "termination", "contact", "terrain", "entity", "command", "observation"]


class LiveAttr:
    """A manager attribute that the reference reads afresh on every step (``self.noise``, ``self.logging_enabled`` …):
    assigning it marks the manager's compiled tables dirty and drops the recorded step."""

    def __init__(self, name: str):
        self.slot = "_live_" + name

    def __get__(self, obj, owner=None):
        return self if obj is None else getattr(obj, self.slot)

    def __set__(self, obj, value):
        had = hasattr(obj, self.slot)
        old = getattr(obj, self.slot, None)
        setattr(obj, self.slot, value)
        if had and old != value:
            obj._live_attr_changed(self.slot[len("_live_"):])


class BaseManager:
    """The base class used to define the interface for all other managers (base.py:16-43)."""

    def __init__(self, env, type: ManagerType, enabled: bool = True):
        self.env = env
        self._enabled = True  # sic: the reference ignores the argument (base.py:28)
        self.type = type
        if hasattr(env, "add_manager"):
            env.add_manager(type, self)

    @property
    def enabled(self) -> bool:
        return self._enabled

    @enabled.setter
    def enabled(self, v: bool):
        changed = bool(v) != bool(self._enabled)
        self._enabled = v
        if changed and hasattr(self.env, "invalidate_trace"):   # (the same value again — a per-step curriculum hook — changes nothing)
            self.env.invalidate_trace()

    def build(self):
        """Called when the scene is built"""

    def step(self):
        """Called when the environment is stepped"""

    def reset(self, envs_idx: list[int] | None = None):
        """One or more environments have been reset"""

    # -- fused-reset protocol (no reference counterpart) ---------------------------------------------
    #: managers whose ``reset`` can be expressed as a section of ``gf_masked_reset`` set this to True and
    #: implement ``_fill_reset``; everything else gets ``reset(envs_idx)`` with a compacted index list.
    _fused_reset = False

    def _live_attr_changed(self, name: str = "") -> None:
        """A plain attribute the reference re-reads on every call was assigned: whatever was compiled from it is stale."""
        if hasattr(self, "_mark_dirty"):
            self._mark_dirty()
        elif hasattr(self, "_dirty"):
            self._dirty = True
        env = getattr(self, "env", None)
        if env is not None and hasattr(env, "invalidate_trace"):
            env.invalidate_trace()

    def _fill_reset(self, args) -> None:
        pass

    def _after_fused_reset(self, mask, mask2) -> None:
        """Follow-up launches that cannot live in the fused call (e.g. command resampling)."""
