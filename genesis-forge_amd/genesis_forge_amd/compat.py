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
    # module paths the reference has and user code imports from (examples/gait_trainer/gait_command_manager.py:10-14)
    from .managers import command as _command

    for alias in ("genesis_forge.managers.command", "genesis_forge.managers.command.command_manager",
                  "genesis_forge.managers.command.velocity_command"):
        sys.modules[alias] = _command
    # … and every other module path of the reference package (deep imports such as
    # ``from genesis_forge.managers.contact.contact_manager import ContactManager``): each resolves to the module of this package that
    # holds the same names (the reference splits managers/action, managers/config, managers/contact and wrappers over several files)
    deep = {
        "managers.action": "managers.action", "managers.action.base": "managers.action",
        "managers.action.position_action_manager": "managers.action", "managers.action.position_within_limits": "managers.action",
        "managers.base": "managers.base",
        "managers.config.config_item": "managers.config", "managers.config.mdp_fn_class": "managers.config",
        "managers.config.params_dict": "managers.config",
        "managers.contact": "managers.contact", "managers.contact.contact_manager": "managers.contact", "managers.contact.config": "managers.contact",
        "managers.entity_manager": "managers.entity_manager", "managers.observation_manager": "managers.observation_manager",
        "managers.reward_manager": "managers.reward_manager", "managers.termination_manager": "managers.termination_manager",
        "managers.terrain_manager": "managers.terrain_manager",
        "wrappers.rsl_rl": "wrappers", "wrappers.skrl": "wrappers", "wrappers.video": "wrappers", "wrappers.wrapper": "wrappers",
    }
    for ref_path, mine in deep.items():
        sys.modules.setdefault("genesis_forge." + ref_path, importlib.import_module(pkg.__name__ + "." + mine))
    sys.modules["genesis_forge.gamepads"] = _gamepads_module()
    if with_genesis_shim and "genesis" not in sys.modules:
        try:
            importlib.import_module("genesis")
            return
        except Exception:
            pass
        shim = make_genesis_shim()
        sys.modules["genesis"] = shim
        sys.modules.update(shim._submodules)


#: extra ``SyntheticScene`` options applied to every ``gs.Scene(...)`` the shim constructs: reference task configs build
#: their scene themselves, so tests and benchmarks steer the stand-in physics (noise levels, contact density, seed) here
SCENE_OVERRIDES: dict = {}


def _gamepads_module() -> types.ModuleType:
    """``genesis_forge.gamepads``: the HID reader is out of scope; the name ``Gamepad`` exists so
    type annotations in user managers resolve, and constructing one says why it cannot work here."""
    m = types.ModuleType("genesis_forge.gamepads")

    class Gamepad:
        def __init__(self, *a, **k):
            raise NotImplementedError("gamepad HID input is outside the manager-step pipeline (SURVEY.md §2 row 19)")

    m.Gamepad = Gamepad
    return m


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
        opts = dict(dt=dt, substeps=substeps, max_collision_pairs=pairs)
        opts.update(SCENE_OVERRIDES)
        return _scene.SyntheticScene(**opts)

    def _init(backend=None, **kw):
        return None

    class _Surfaces:
        Default = Rough = Smooth = Plastic = _Opt

    class _Textures:
        ImageTexture = ColorTexture = _Opt

    m.surfaces = _Surfaces
    m.textures = _Textures
    m.__path__ = []  # a package, so ``from genesis.engine.entities import RigidEntity`` resolves
    eng = types.ModuleType("genesis.engine")
    eng.__path__ = []
    ents = types.ModuleType("genesis.engine.entities")
    ents.RigidEntity = _scene.SyntheticEntity
    eng.entities = ents
    m.engine = eng
    m._submodules = {"genesis.engine": eng, "genesis.engine.entities": ents}
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
