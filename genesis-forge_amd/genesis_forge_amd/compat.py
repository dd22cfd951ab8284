"""Module aliasing so reference task configs (``from genesis_forge.managers import …``,
``import genesis as gs``) run against this package unchanged."""
from __future__ import annotations

import importlib
import sys
import types


def install(pkg, with_genesis_shim: bool = True) -> None:
    names = ["", ".managers", ".mdp", ".mdp.rewards", ".mdp.terminations", ".mdp.observations", ".mdp.reset", ".utils",
             ".wrappers", ".genesis_env", ".managed_env", ".managers.config"]
    for n in names:
        mod = importlib.import_module(pkg.__name__ + n)
        sys.modules["genesis_forge" + n] = mod
    if with_genesis_shim and "genesis" not in sys.modules:
        try:
            importlib.import_module("genesis")
            return
        except Exception:
            pass
        sys.modules["genesis"] = make_genesis_shim()


def make_genesis_shim() -> types.ModuleType:
    """A ``genesis`` look-alike: ``gs.init``, ``gs.device``, ``gs.Scene`` (synthetic), ``gs.morphs``, ``gs.options``."""
    from . import gs as _gs
    from . import scene as _scene

    m = types.ModuleType("genesis")
    m.__dict__.update(tc_float=_gs.tc_float, tc_int=_gs.tc_int, tc_bool=_gs.tc_bool, JOINT_TYPE=_gs.JOINT_TYPE,
                      morphs=_scene.morphs, gpu="gpu", cpu="cpu")

    class _Opt:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class _Options:
        SimOptions = ViewerOptions = VisOptions = RigidOptions = _Opt

    class _Solver:
        Newton = "newton"

    def _scene_factory(show_viewer=False, sim_options=None, rigid_options=None, **kw):
        dt = getattr(sim_options, "dt", None) or getattr(rigid_options, "dt", None) or 0.02
        substeps = getattr(sim_options, "substeps", 1)
        pairs = getattr(rigid_options, "max_collision_pairs", 0) or 0
        return _scene.SyntheticScene(dt=dt, substeps=substeps, max_collision_pairs=pairs)

    def _init(backend=None, **kw):
        return None

    m.options = _Options
    m.constraint_solver = _Solver
    m.Scene = _scene_factory
    m.init = _init

    def __getattr__(name):
        if name == "device":
            return _gs.device
        raise AttributeError(name)

    m.__getattr__ = __getattr__
    return m
