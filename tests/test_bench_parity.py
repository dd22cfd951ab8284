"""Every kernel that gets TIMED is compared with the oracle at the size it is timed at.

bench.py times BASELINE.json's Go2 config on the recorded step — action kernel, scene tick, and the fused post-physics launch
running a STATIC PROGRAM (gf::post_ws_kernel<ProgGo2CommandDirection>) — at 65 536 envs (value), 4 096 / 16 384 (sweep) and
1 048 576 (roofline_hbm); tools/bench_configs.py times the other BASELINE configs.  The other GPU parity tests reach the fused
kernels only through "recorded == phase-by-phase" (a HIP-vs-HIP comparison) or run configs whose observation noise keeps them
on the table interpreter.  Here the recorded, fused HIP env and a PHASE-BY-PHASE oracle env (no recording: the plain
restatement of the reference's phase order) are stepped side by side in Philox mode on identical inputs, and every step's
outputs and manager state are compared: masks / counters bit-exact, floats within 1e-5 (BASELINE.json).

Sizes above one tile matter for what the small fixtures cannot reach: 32-bit row offsets (N * row_bytes close to 4 GiB at
1 048 576 + 37 envs with history), partial tail tiles, the statistics shards (workgroup % 64) and the XCD-aligned action tiles.
"""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5
WARM_STEPS = 200

# (config of genesis_forge_amd.tasks.BASELINE_CONFIGS or a stress config, num_envs, variant, steps, expected post-physics kernel)
#   variant "bench": exactly the timed workload (20 s episodes: resets come from terminations only), compared over `steps`
#                    steps AFTER 200 uncompared lockstep steps — the scene needs that long to reach the steady state bench.py
#                    times (about 0.2 % of the envs falling over per step)
#   variant "short": same structure (same kernel), 0.3 s episodes and 0.1 s command resampling, so time-outs, the
#                    episode-length jitter, command.step resamples and the RewardManager's reset all fire at this size
CASES = [
    ("go2_cmd", 64, "short", 24, "go2_command_direction"),
    ("go2_cmd", 4096, "bench", 20, "go2_command_direction"),
    ("go2_cmd", 16384, "short", 20, "go2_command_direction"),
    ("go2_cmd", 65536, "bench", 20, "go2_command_direction"),
    ("go2_cmd", 65536, "short", 24, "go2_command_direction"),
    ("go2_cmd", 1048576 + 37, "short", 18, "go2_command_direction"),
    ("simple", 64, "short", 24, "go2_simple"),
    ("simple", 4096, "bench", 20, "go2_simple"),
    ("contacts", 4096, "bench", 20, "go2_contacts"),
    ("contacts", 4096, "short", 24, "go2_contacts"),
    ("rough_terrain", 16384, "bench", 20, "go2_rough_terrain"),
    ("rough_terrain", 16384, "short", 24, "go2_rough_terrain"),
    ("humanoid", 8192, "bench", 20, "berkeley_humanoid"),
    ("humanoid", 8192, "short", 24, "berkeley_humanoid"),
    ("gait", 8192, "bench", 20, None),
    ("gait", 65536, "short", 20, None),
    ("humanoid28", 8192, "short", 24, "humanoid28_stress"),   # BASELINE config 4 as stated ("~28-DOF"): synthetic 28-DOF humanoid, static program 7
    ("humanoid28", 8192, "interp", 24, "interpreter"),        # … and the 28-DOF table interpreter every other 28-DOF config runs (GF_OPT_POST_VARIANT = 1)
    ("go2_cmd", 65536, "interp", 18, "interpreter"),          # … and the 12-DOF interpreter at the benchmark size
    ("go2_user", 65536, "jit", 20, "jit_"),                   # a structure the library was not built with, on its run-time compiled program
    ("gait_override_8192", 8192, "short", 24, "go2_gait_trainer_front"),   # the gait example as shipped (reset() override): the two split launches
]


def _make(name, n, variant):
    from genesis_forge_amd import tasks

    if name == "humanoid28":
        return tasks.HumanoidGaitLikeEnv(num_envs=n, dofs=28)
    kw = {} if variant == "bench" else {"max_episode_length_s": 0.3}   # ("short" / "interp": 0.3 s episodes)
    env = tasks.BASELINE_CONFIGS[name][1](n, **kw)
    return env


class _Side:
    """Makes one (device, backend) pair the process-wide current one while an env of that side is built or stepped."""

    def __init__(self, dev, backend):
        self.dev, self.backend = dev, backend

    def __enter__(self):
        from genesis_forge_amd import _native as nat
        from genesis_forge_amd import gs

        self.old = (gs.device, nat._backend)
        gs.set_device(self.dev)
        nat.set_backend(self.backend)

    def __exit__(self, *exc):
        from genesis_forge_amd import _native as nat
        from genesis_forge_amd import gs

        gs.device = self.old[0]
        nat.set_backend(self.old[1])


def _state(env, out):
    obs, rew, term, trunc, extras = out
    exact = {"terminated": term, "truncated": trunc, "episode_length": env.episode_length, "max_episode_length": env.max_episode_length}
    close = {"obs": obs, "reward": rew, "pos": env.robot.get_pos(), "quat": env.robot.get_quat(),
             "dof_pos": env.managers["action"].get_dofs_position(), "targets": env.managers["action"].get_actions(),
             "episode_sums": env.managers["reward"]._episode_sums, "episode_seconds": env.managers["reward"]._episode_seconds,
             "env_actions": env.actions, "env_last_actions": env.last_actions}
    for k, v in extras["observations"].items():
        if k != "policy":
            close["obs_" + k] = v
    for i, m in enumerate(env.managers["command"]):
        close[f"command{i}"] = m.command
    for i, m in enumerate(env.managers["contact"]):
        close[f"contacts{i}"] = m.contacts
        if m._track_air_time:
            close[f"air{i}"] = torch.stack([m.last_air_time, m.current_air_time, m.last_contact_time, m.current_contact_time])
    g = getattr(env, "gait_command_manager", None)
    if g is not None:
        exact["gait_selected"] = g._gait_selected
        close["gait_state"] = g._state
    return exact, close, {k: float(v) for k, v in extras["episode"].items()}


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,variant,steps,program", CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}" for c in CASES])
def test_timed_workload_hip_equals_oracle(hip_backend, oracle_lib_path, name, n, variant, steps, program):
    from genesis_forge_amd import _native as nat
    from oracle_backend import OracleBackend

    hip, cpu = _Side("cuda:0", hip_backend), _Side("cpu", OracleBackend(oracle_lib_path))
    if variant == "jit":   # GF_JIT=sync: the config's own program is compiled (≈ 5 s, cached) and registered when the step is recorded
        os.environ["GF_JIT"] = "sync"
        try:
            _timed_workload(hip_backend, hip, cpu, name, n, "short", steps, program)
        finally:
            del os.environ["GF_JIT"]
        return
    if variant == "interp":   # keep the config off its static program: the table interpreter of the fused launch at this size
        hip_backend.set_option(nat.GF_OPT_POST_VARIANT, 1)
        try:
            _timed_workload(hip_backend, hip, cpu, name, n, "short", steps, program)
        finally:
            hip_backend.set_option(nat.GF_OPT_POST_VARIANT, 2)
        return
    _timed_workload(hip_backend, hip, cpu, name, n, variant, steps, program)


def _timed_workload(hip_backend, hip, cpu, name, n, variant, steps, program):
    envs = {}
    for key, side in (("hip", hip), ("cpu", cpu)):
        with side:
            env = _make(name, n, variant)
            if key == "cpu":
                env.trace_enabled = False     # the oracle side is the plain phase-by-phase restatement of the reference's order
            env.build()
            if variant == "short":
                for m in env.managers["command"]:
                    m.resample_time_sec = 0.1 if m is env.managers["command"][0] else 0.14
            env.seed(4242)
            env.reset()
            envs[key] = env
    d = envs["hip"].action_space.shape[0]
    g = torch.Generator().manual_seed(11)
    resets = 0
    for t in range(WARM_STEPS if variant == "bench" else 0):
        act = torch.randn(n, d, generator=g)
        with hip:
            envs["hip"].step(act.to("cuda:0"))
        with cpu:
            envs["cpu"].step(act)
    for t in range(steps):
        act = torch.randn(n, d, generator=g)
        with hip:
            a = _state(envs["hip"], envs["hip"].step(act.to("cuda:0")))
        with cpu:
            b = _state(envs["cpu"], envs["cpu"].step(act))
        for k in a[0]:
            assert torch.equal(a[0][k].cpu(), b[0][k]), f"{k} differs at step {t}"
        for k in a[1]:
            x, y = a[1][k].cpu(), b[1][k]
            assert x.shape == y.shape, k
            assert torch.allclose(x, y, atol=TOL, rtol=0), f"{k} differs at step {t}: max |d| = {(x - y).abs().max()}"
        assert set(a[2]) == set(b[2]), f"log keys differ at step {t}: {sorted(a[2])} vs {sorted(b[2])}"
        for k in a[2]:
            assert abs(a[2][k] - b[2][k]) <= 1e-5 + 1e-5 * abs(b[2][k]), (t, k, a[2][k], b[2][k])
        resets += int(b[0]["terminated"].sum() + b[0]["truncated"].sum())
    tr = envs["hip"]._trace
    assert tr is not None, "the timed step is the recorded one"
    if program is not None:
        assert tr.post_refs is not None, "the post-physics phases of this config run as the fused launch"
        what = hip_backend.post_describe(tr.post_refs)
        assert (f"({program}" if program.endswith("_") else f"({program})") in what.split(":")[0], what
        if name == "gait_override_8192":
            assert tr.tail_seg.get("obs", {}).get("fused_obs") and "(go2_gait_trainer_obs)" in hip_backend.post_describe(tr._tail_refs).split(":")[0]
    assert envs["cpu"]._trace is None
    if variant == "short" or (name == "go2_cmd" and n >= 4096):   # (rough terrain's 30-degree limit never fires at the bench noise level)
        assert resets > 0, "the trajectory must reset envs"


FOLD_CASES = [  # (fold = 2: "whenever possible" — the default keeps multi-pass tiles of small launches on two launches, measured faster)
    ("contacts", 4096), ("rough_terrain", 1000), ("rough_terrain", 16384), ("humanoid", 130), ("humanoid", 8192), ("gait", 8192), ("gait", 65536),
              ("gait_override_8192", 8192), ("humanoid28", 1000), ("humanoid28-interp", 257)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,n", FOLD_CASES, ids=[f"{c[0]}-{c[1]}" for c in FOLD_CASES])
def test_contact_phase_folded_into_the_post_launch_equals_a_launch_of_its_own(hip_backend, name, n):
    """SURVEY.md §8f-1 ∘ §8f-2: gf_run_ops hands the contact ops in front of the fused post-physics op to that launch
    (gf_post_physics_step_contacts; managed_env.py:294-326 runs the phases back to back per env).  Same inputs, GF_OPT_FOLD_CONTACT
    on / off: every output and every manager buffer bit for bit, and in the folded run the contact kernel is never launched."""
    from genesis_forge_amd import _native as nat

    interp = name.endswith("-interp")
    cfg = name.split("-")[0]
    runs = {}
    for fold in (2, 0):
        hip_backend.set_option(nat.GF_OPT_FOLD_CONTACT, fold)
        if interp:
            hip_backend.set_option(nat.GF_OPT_POST_VARIANT, 1)
        try:
            env = _make(cfg, n, "short")
            env.build()
            for m in env.managers["command"]:
                m.resample_time_sec = 0.1 if m is env.managers["command"][0] else 0.14
            env.seed(99)
            env.reset()
            d = env.action_space.shape[0]
            g = torch.Generator().manual_seed(5)
            rows = []
            for t in range(30):
                if t == 6:   # the step is recorded by now: count the contact launches of the replayed steps
                    assert env._trace is not None
                    hip_backend.profile_begin(nat.GF_PHASE_CONTACT, 64)
                act = torch.randn(n, d, generator=g).to("cuda:0")
                ex, cl, log = _state(env, env.step(act))
                extra = {}
                for i, m in enumerate(env.managers["contact"]):
                    extra[f"pos{i}"] = m.contact_positions
                    for nm in ("link_vel", "link_pos"):
                        if getattr(m, "_has_" + nm, False):
                            extra[f"{nm}{i}"] = getattr(m, nm)
                rows.append(({k: v.clone() for k, v in {**ex, **cl, **extra}.items()}, log))
            _ms, samples = hip_backend.profile_end()
            runs[fold] = (rows, samples)
        finally:
            hip_backend.set_option(nat.GF_OPT_FOLD_CONTACT, 0 if os.environ.get("GF_NO_CONTACT_FOLD") else 1)
            hip_backend.set_option(nat.GF_OPT_POST_VARIANT, 2)
    assert runs[2][1] == 0, "folded: the recorded step launches no contact kernel"
    assert runs[0][1] == 24, "a launch of its own per step when the fold is switched off"
    for t, ((a, la), (b, lb)) in enumerate(zip(runs[2][0], runs[0][0])):
        assert set(a) == set(b)
        for k in a:
            assert torch.equal(a[k], b[k]), f"{k} differs at step {t}"
        assert la == lb, t


def test_every_timed_config_has_a_parity_case():
    """`configs_untested` is empty: every entry of the table bench.py / tools/bench_configs.py time has a case at its size."""
    from genesis_forge_amd.tasks import BASELINE_CONFIGS

    covered = {(c[0], c[1]) for c in CASES}
    alias = {"go2_cmd_65536": "go2_cmd", "gait_8192": "gait"}
    for name, (size, _make_env) in BASELINE_CONFIGS.items():
        assert (alias.get(name, name), size) in covered, f"{name} @ {size} is timed but never compared with the oracle at that size"
    for n in (4096, 16384, 65536, 1048576 + 37):   # bench.py: sweep, value, roofline_hbm
        assert ("go2_cmd", n) in covered
