"""
genesis_forge_amd — MI355X-native manager step pipeline behind genesis-forge's
``ManagedEnvironment`` / ``Manager`` / ``mdp.*`` plugin API.

The per-tick manager work that runs across ``num_envs`` executes as hand-written HIP kernels
(``libgf_step.so``, C ABI in ``include/gf_step.h``); this package is the host-side mirror of the
reference's operator interface.  ``install_as_genesis_forge()`` registers it under the module name
``genesis_forge`` so task configs written for the reference import unchanged.
"""
from .genesis_env import GenesisEnv, EnvMode
from .managed_env import ManagedEnvironment

__all__ = ["GenesisEnv", "ManagedEnvironment", "EnvMode", "install_as_genesis_forge"]
__version__ = "0.1.0"


def install_as_genesis_forge(with_genesis_shim: bool = True) -> None:
    """Alias this package (and its submodules) as ``genesis_forge`` in ``sys.modules``; when the real
    ``genesis`` package is absent and ``with_genesis_shim`` is set, also register the synthetic
    ``genesis`` stand-in so ``import genesis as gs`` in a task config resolves."""
    import importlib
    import sys

    from . import compat

    compat.install(sys.modules[__name__], with_genesis_shim)
