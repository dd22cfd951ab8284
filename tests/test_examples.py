"""The six task configs the reference ships (examples/*/environment.py), end to end against the REFERENCE.

tests/golden/traj_ex_<name>.npz were recorded by tools/gen_golden.py from the reference package running its own example
file (unchanged) on the synthetic scene: every step's observations (policy and critic), reward, done masks, episode
counters, commands, base pose after resets, gait-manager state and logged scalars.

* ``test_reference_example_file_drops_in``: the reference's example FILE itself is executed against this package under the
  ``genesis_forge`` / ``genesis`` aliases — "the repo's Go2 and humanoid task configs drop in unchanged" (BASELINE.json).
  Runs only where /root/reference exists (the build container); CPU, oracle as the compute backend.
* ``test_restated_config_*``: tests/envs.py restates the same six configs (the reference cannot travel to the GPU box) and
  must reproduce the same fixtures on the CPU oracle and on the HIP kernels (masks bit-exact, floats <= 1e-5).
"""
import os
import sys

import numpy as np
import pytest
import torch

import example_cases
import helpers

EXAMPLES = list(example_cases.CASES)
RESTATED = EXAMPLES  # gait_trainer runs on the native GaitCommandManager
REF_EXAMPLES = "/root/reference/examples"


def _load_reference_example(name):
    """Execute the reference's example file against this package (aliased as genesis_forge / genesis)."""
    import importlib.util

    import genesis_forge_amd

    sys.dont_write_bytecode = True  # never leave __pycache__ in /root/reference
    genesis_forge_amd.install_as_genesis_forge()
    d = os.path.join(REF_EXAMPLES, name)
    sys.path.insert(0, d)
    for stale in ("environment", "gait_command_manager"):
        sys.modules.pop(stale, None)
    try:
        spec = importlib.util.spec_from_file_location("environment", os.path.join(d, "environment.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["environment"] = mod
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(d)
    cls = [v for v in vars(mod).values() if isinstance(v, type) and issubclass(v, genesis_forge_amd.ManagedEnvironment)
           and v is not genesis_forge_amd.ManagedEnvironment]
    assert len(cls) == 1
    return cls[0], mod


def _unalias():
    for k in [k for k in sys.modules if k == "genesis" or k.startswith("genesis.") or k == "genesis_forge" or k.startswith("genesis_forge.")
              or k in ("environment", "gait_command_manager")]:
        del sys.modules[k]


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES), reason="the reference's example files exist only in the build container")
@pytest.mark.parametrize("name", EXAMPLES)
def test_reference_example_file_drops_in(oracle_backend, name):
    from genesis_forge_amd import compat

    case = example_cases.CASES[name]
    fix = helpers.load(f"traj_ex_{name}")
    try:
        compat.SCENE_OVERRIDES.clear()
        compat.SCENE_OVERRIDES.update(case["scene"])
        cls, mod = _load_reference_example(name)
        env = cls(num_envs=case["n"], max_episode_length_s=case["episode_s"])
        compat.SCENE_OVERRIDES.clear()
        user_gait = getattr(mod, "GaitCommandManager", None)
        res = helpers.replay_example(fix, case, env, "cpu", user_gait_cls=user_gait)
        helpers.compare_example(fix, res)
    finally:
        compat.SCENE_OVERRIDES.clear()
        _unalias()


@pytest.mark.parametrize("name", RESTATED)
def test_restated_config_cpu_oracle(oracle_backend, name):
    import envs

    case = example_cases.CASES[name]
    fix = helpers.load(f"traj_ex_{name}")
    env = envs.make_example(name, case)
    res = helpers.replay_example(fix, case, env, "cpu")
    helpers.compare_example(fix, res)


@pytest.mark.gpu
@pytest.mark.parametrize("name", RESTATED)
def test_restated_config_hip(hip_backend, name):
    import envs

    case = example_cases.CASES[name]
    fix = helpers.load(f"traj_ex_{name}")
    env = envs.make_example(name, case)
    res = helpers.replay_example(fix, case, env, "cuda")
    helpers.compare_example(fix, res)
