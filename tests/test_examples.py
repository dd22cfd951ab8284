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

CASE_KEYS = [k for k, c in example_cases.CASES.items() if not c.get("compact")]   # the six examples at n = 8 plus the multi-tile cases (<example>_n<envs>)
COMPACT_KEYS = [k for k, c in example_cases.CASES.items() if c.get("compact")]     # … and at a timed size, as a compact fixture (helpers.compact_example)
EXAMPLES = [k for k in CASE_KEYS if example_cases.example_of(k) == k]
RESTATED = CASE_KEYS  # gait_trainer runs on the native GaitCommandManager
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
@pytest.mark.parametrize("name", CASE_KEYS)
def test_reference_example_file_drops_in(oracle_backend, name):
    from genesis_forge_amd import compat

    case = example_cases.CASES[name]
    fix = helpers.load(f"traj_ex_{name}")
    try:
        compat.SCENE_OVERRIDES.clear()
        compat.SCENE_OVERRIDES.update(case["scene"])
        cls, mod = _load_reference_example(example_cases.example_of(name))
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
    env = envs.make_example(example_cases.example_of(name), case)
    res = helpers.replay_example(fix, case, env, "cpu")
    helpers.compare_example(fix, res)


@pytest.mark.gpu
@pytest.mark.parametrize("name", RESTATED)
def test_restated_config_hip(hip_backend, name):
    import envs

    case = example_cases.CASES[name]
    fix = helpers.load(f"traj_ex_{name}")
    env = envs.make_example(example_cases.example_of(name), case)
    res = helpers.replay_example(fix, case, env, "cuda")
    helpers.compare_example(fix, res)


def _compact_case(name, dev):
    """The restated config at a timed size against the reference's own example file run at that size: per-step sums over all envs +
    a strided 64-env sample of every recorded field (outputs, counters, commands, gait-manager state, logged scalars)."""
    import envs

    case = example_cases.CASES[name]
    fix = helpers.load(f"traj_ex_{name}")
    assert int(fix["compact"]) == 1 and int(fix["n"]) == case["n"]
    env = envs.make_example(example_cases.example_of(name), case)
    res = helpers.replay_example({"seed": fix["seed"], "actions": helpers.example_actions(case)}, case, env, dev, compact=True)
    logs = res.pop("logs")
    helpers.compare_compact_example(fix, res, logs)


@pytest.mark.parametrize("name", COMPACT_KEYS)
def test_restated_config_at_size_cpu_oracle(oracle_backend, name):
    _compact_case(name, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", COMPACT_KEYS)
def test_restated_config_at_size_hip(hip_backend, name):
    _compact_case(name, "cuda")


def _philox_run(name, dev, trace, n, steps=45):
    """The restated config in Philox mode (no parity draws, so the step can be recorded and fused)."""
    import envs

    case = dict(example_cases.CASES[name], n=n, episode_s=1.0)
    env = envs.make_example(name, case) if name != "gait_trainer" else envs.Go2GaitTrainingEnv(
        num_envs=n, max_episode_length_s=1.0, scene_kwargs=dict(case["scene"]))
    env.trace_enabled = trace
    env.build()
    env.seed(17)
    for attr, sec in case["resample"].items():
        getattr(env, attr).resample_time_sec = sec
    env.reset()
    g = torch.Generator().manual_seed(5)
    d = env.action_space.shape[0]
    out = []
    for _ in range(steps):
        o, r, te, tr, ex = env.step(torch.randn(n, d, generator=g).to(dev))
        others = [v.cpu().clone() for k, v in ex["observations"].items() if k != "policy"]
        out.append(([o.cpu().clone(), r.cpu().clone(), te.cpu().clone(), tr.cpu().clone()] + others,
                    {k: float(v) for k, v in ex["episode"].items()}))
    return out, env


def _override_run(dev, trace, n, steps=45):
    """The gait task WITH the example's reset() override (examples/gait_trainer/environment.py:347-352: curriculum update after
    every reset) and, on top, a counter the override bumps — user code in reset() must run exactly as often as in an ordinary step."""
    import envs

    case = dict(example_cases.CASES["gait_trainer"], n=n, episode_s=1.0)

    class Env(envs.Go2GaitTrainingCurriculumEnv):
        user_resets = 0
        user_reset_envs = 0

        def reset(self, envs_idx=None):
            out = envs.Go2GaitTrainingCurriculumEnv.reset(self, envs_idx)
            if envs_idx is not None:
                self.user_resets += 1
                self.user_reset_envs += int(len(envs_idx))
            return out

    env = Env(num_envs=n, max_episode_length_s=1.0, scene_kwargs=dict(case["scene"]))
    env.trace_enabled = trace
    env.build()
    env.seed(17)
    for attr, sec in case["resample"].items():
        getattr(env, attr).resample_time_sec = sec
    env.reset()
    g = torch.Generator().manual_seed(5)
    d = env.action_space.shape[0]
    out = []
    for _ in range(steps):
        o, r, te, tr, ex = env.step(torch.randn(n, d, generator=g).to(dev))
        others = [v.cpu().clone() for k, v in ex["observations"].items() if k != "policy"]
        out.append(([o.cpu().clone(), r.cpu().clone(), te.cpu().clone(), tr.cpu().clone()] + others,
                    {k: float(v) for k, v in ex["episode"].items()}))
    return out, env


def _check_override(dev, n):
    a, e0 = _override_run(dev, False, n)
    b, env = _override_run(dev, True, n)
    tr = env._trace
    assert tr is not None and tr.tail_python, "an env that overrides reset() is recorded up to the reset; the user's reset() stays Python"
    assert sorted(tr.tail_seg) == ["obs", "reset"], "what super().reset(ids) and get_observations() launch replays natively, part by part"
    assert tr.tail_seg["obs"]["fused_obs"], "both observation managers of the tail run as ONE launch of the fused kernel's observation waves (GF_POST_OBSERVE_ONLY)"
    assert (env.user_resets, env.user_reset_envs) == (e0.user_resets, e0.user_reset_envs) and env.user_resets > 0
    _same_runs(a, b)


def test_reset_override_env_is_recorded_up_to_the_reset_cpu(oracle_backend):
    _check_override("cpu", 70)


@pytest.mark.gpu
def test_reset_override_env_is_recorded_up_to_the_reset_hip(hip_backend):
    _check_override("cuda", 1000)


def _same_runs(a, b):
    for t, ((x, lx), (y, ly)) in enumerate(zip(a, b)):
        assert len(x) == len(y)
        for k, (u, v) in enumerate(zip(x, y)):
            assert torch.equal(u, v), f"output {k} differs at step {t}"
        assert lx == ly, f"log differs at step {t}: {lx} vs {ly}"


@pytest.mark.parametrize("name", EXAMPLES)
def test_recorded_step_equals_ordinary_cpu(oracle_backend, name):
    a, _ = _philox_run(name, "cpu", False, 70)
    b, env = _philox_run(name, "cpu", True, 70)
    assert env._trace is not None, "the config's step should be recorded"
    _same_runs(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("name", EXAMPLES)
def test_recorded_step_equals_ordinary_hip(hip_backend, name):
    """Recorded + fused (or chained) launches vs the phase-by-phase path, bit for bit, for every shipped task structure."""
    a, _ = _philox_run(name, "cuda", False, 1000)
    b, env = _philox_run(name, "cuda", True, 1000)
    assert env._trace is not None, "the config's step should be recorded"
    if name != "gait_trainer":
        assert env._trace.post_refs is not None, "the post-physics phases of this config should run as the fused launch"
    _same_runs(a, b)


@pytest.mark.skipif(not os.path.isdir(REF_EXAMPLES), reason="the reference's example files exist only in the build container")
@pytest.mark.parametrize("name", EXAMPLES)
def test_reference_example_file_takes_the_fast_path(oracle_backend, name):
    """The unchanged example files are not only correct on this package, they get the recorded step and the fused post-physics
    launch (observation lambdas are fused by provenance).  gait_trainer — a user-defined CommandManager class with its own
    step() / reset(), an env reset() override — is recorded too (round 3): the user manager's step() is replayed as user code
    between the native phases, the reset() override keeps the tail Python."""
    from genesis_forge_amd import compat

    case = example_cases.CASES[name]
    try:
        compat.SCENE_OVERRIDES.clear()
        compat.SCENE_OVERRIDES.update(case["scene"])
        cls, _mod = _load_reference_example(name)
        env = cls(num_envs=64)
        compat.SCENE_OVERRIDES.clear()
        env.build()
        env.reset()
        d = env.action_space.shape[0]
        for _ in range(6):
            env.step(torch.zeros(64, d))
        if name == "gait_trainer":
            tr = env._trace
            assert tr is not None and tr.tail_python and tr.post_refs is None
            assert len(tr.py_marks) >= 1, "the user manager's step() is a split of the recording"
        else:
            assert env._trace is not None and env._trace.post_refs is not None
            assert env._trace.n_ops <= 5
    finally:
        compat.SCENE_OVERRIDES.clear()
        _unalias()
