"""
Manager config plumbing: ``ParamsDict``, ``MdpFnClass`` / ``ResetMdpFnClass`` and the ``ConfigItem`` family.

API mirror of genesis_forge/managers/config/ (params_dict.py:4-22, mdp_fn_class.py:5-46,
config_item.py:9-114): cfg dict entries ``{fn, params, weight | time_out | scale, noise}`` are wrapped
so that class-style fns are instantiated at build and rebuilt when ``params`` are mutated, and
``weight`` / ``params`` stay live-mutable (curricula).  One addition: every mutation also marks the
owning manager's compiled native term table dirty, because here weights and params are uploaded
into a term table instead of being re-read by a Python loop each step (SURVEY.md §8b "Mutability").
"""
from __future__ import annotations

import inspect
from typing import Callable, Optional


def _same_plain(old, new) -> bool:
    """An assignment that changes nothing: the same immutable scalar again.  The reference's own curriculum recipe assigns its params
    on EVERY step, mostly with the value they already have (docs/guide/managers/termination.md "Curriculum-Based Termination": step() →
    update_curriculum() → ``term_cfg[...].params["angle_limit"] = angle_limit``); there that costs nothing, here a change drops the
    compiled term table and the recorded step.  Only plain immutable values qualify — re-assigning a mutable value (a list, a tensor) is
    the documented way to announce an in-place edit and must keep marking the table dirty."""
    if old is None or new is None:
        return old is new
    if isinstance(old, bool) or isinstance(new, bool):
        return isinstance(old, bool) and isinstance(new, bool) and old == new
    if isinstance(old, (int, float)) and isinstance(new, (int, float)):
        return old == new   # (10 and 10.0 compile to the same table entry; NaN never equals itself: treated as a change)
    if isinstance(old, str) and isinstance(new, str):
        return old == new
    return False


class ParamsDict(dict):
    """dict that reports every mutation to ``on_change``.

    The reference hooks item assignment / deletion only (params_dict.py:4-22) and gets away with it because its term loops
    re-read ``**params`` every step, so ``params.update(...)``, ``pop``, ``setdefault``, ``clear`` and ``|=`` take effect
    there too.  Here params are compiled into a native term table (and frozen into a recorded step), so every mutating method
    must mark the table dirty.  What no dict can see — editing a mutable VALUE in place (a list appended to, a tensor written
    through a view the table copied) — still needs an explicit ``params[key] = params[key]``."""

    def __init__(self, params: dict, on_change: Callable[[], None]):
        super().__init__(params)
        self._on_change = on_change

    def __setitem__(self, key, value):
        same = key in self and _same_plain(self[key], value)
        super().__setitem__(key, value)
        if not same:
            self._on_change()

    def __delitem__(self, key):
        super().__delitem__(key)
        self._on_change()

    def update(self, *args, **kwargs):
        new = dict(*args, **kwargs)
        same = all(k in self and _same_plain(self[k], v) for k, v in new.items())
        super().update(new)
        if not same:
            self._on_change()

    def pop(self, *args):
        out = super().pop(*args)
        self._on_change()
        return out

    def popitem(self):
        out = super().popitem()
        self._on_change()
        return out

    def setdefault(self, key, default=None):
        had = key in self
        out = super().setdefault(key, default)
        if not had:
            self._on_change()
        return out

    def clear(self):
        super().clear()
        self._on_change()

    def __ior__(self, other):
        self.update(other)
        return self


class MdpFnClass:
    """Callable class usable wherever an MDP function is; ``build`` runs at env build and on param change."""

    def __init__(self, env):
        self.env = env

    def build(self):
        pass

    def __call__(self, env, envs_idx=None):
        pass


class ResetMdpFnClass(MdpFnClass):
    """MDP function class for EntityManager ``on_reset`` entries: called as ``fn(env, entity, envs_idx, **params)``."""

    def __init__(self, env, entity):
        self.env = env

    def __call__(self, env, entity, envs_idx):
        pass


class ConfigItem:
    """One cfg entry.  ``on_dirty`` is invoked whenever something the native term table depends on changes."""

    def __init__(self, cfg: dict, env, on_dirty: Optional[Callable[[], None]] = None):
        self._env = env
        self._kwargs: dict = {}
        self._on_dirty = on_dirty
        self._soft_ok = self._what_ok = False
        if on_dirty is not None:
            try:
                pars = inspect.signature(on_dirty).parameters
                self._soft_ok, self._what_ok = "soft" in pars, "what" in pars
            except (TypeError, ValueError):
                pass
        self._cfg = cfg
        self._fn = cfg["fn"]
        self._params = ParamsDict(cfg.get("params", {}) or {}, self._rebuild)
        self._is_class = inspect.isclass(cfg["fn"])
        self._initialized = not self._is_class

    def _dirty(self, soft: bool = False, what: Optional[str] = None):
        """``soft``: only NUMBERS of the compiled term table can have changed (a param value, a weight): a manager that knows how may
        refresh the table a recorded step uses in place instead of dropping the recording (RewardManager / TerminationManager).
        ``what``: the field that was assigned ("_weight" …), for managers that can refresh less than the whole table."""
        if self._on_dirty is not None:
            if soft and self._soft_ok:
                if self._what_ok:
                    self._on_dirty(soft=True, what=what)
                else:
                    self._on_dirty(soft=True)
            else:
                self._on_dirty()

    @property
    def fn(self):
        return self._fn

    @property
    def params(self):
        return self._params

    @params.setter
    def params(self, params: dict):
        self._params = ParamsDict(params.copy(), self._rebuild)
        self._dirty()
        if self._is_class:
            self._rebuild()

    def build(self, **kwargs):
        """Instantiate a class-style fn; ``kwargs`` (e.g. ``entity=``) are passed to its ctor and to every call."""
        self._kwargs = kwargs
        if self._is_class:
            self._init_fn_class()

    def execute(self, envs_idx):
        self._fn(self._env, **self._kwargs, envs_idx=envs_idx, **self._params)

    def _init_fn_class(self):
        if self._initialized:
            return
        instance = self._fn(self._env, **self._kwargs, **(self._cfg.get("params", {}) or {}))
        instance.build()
        self._fn = instance
        # the reference leaves _initialized False here (config_item.py:77) so a later param change
        # would call the *instance* as a constructor; instances are kept and simply re-built instead.
        self._initialized = True

    def _rebuild(self):
        self._dirty(soft=not self._is_class)   # (a class-style fn is re-built from its params: anything may change)
        if self._is_class and self._initialized and hasattr(self._fn, "build"):
            self._fn.build()


class _Field:
    """Attribute whose assignment marks the item dirty."""

    def __init__(self, name):
        self.name = "_" + name

    def __get__(self, obj, owner=None):
        return self if obj is None else getattr(obj, self.name)

    def __set__(self, obj, value):
        same = _same_plain(getattr(obj, self.name, None), value) and hasattr(obj, self.name)
        setattr(obj, self.name, value)
        if not same:
            obj._dirty(soft=self.name in ("_weight", "_scale", "_noise"), what=self.name)   # numbers of a compiled table; `time_out` is structure


class TerminationConfigItem(ConfigItem):
    time_out = _Field("time_out")

    def __init__(self, cfg: dict, env, on_dirty=None):
        super().__init__(cfg, env, on_dirty)
        self._time_out = cfg.get("time_out", False)


class RewardConfigItem(ConfigItem):
    weight = _Field("weight")

    def __init__(self, cfg: dict, env, on_dirty=None):
        super().__init__(cfg, env, on_dirty)
        self._weight = cfg.get("weight", 0.0)


class ObservationConfigItem(ConfigItem):
    scale = _Field("scale")
    noise = _Field("noise")

    def __init__(self, cfg: dict, env, on_dirty=None):
        super().__init__(cfg, env, on_dirty)
        self._scale = cfg.get("scale", 1.0)
        self._noise = cfg.get("noise", None)


__all__ = ["ConfigItem", "MdpFnClass", "ParamsDict", "ResetMdpFnClass", "RewardConfigItem", "TerminationConfigItem",
           "ObservationConfigItem"]
