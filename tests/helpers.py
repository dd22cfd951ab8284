"""Shared helpers for the parity tests: fixture loading, parity-mode draws, trajectory replay."""
import ast
import os

import numpy as np
import torch

import philox

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FLOAT_TOL = 1e-5  # BASELINE.json north_star: within 1e-5 abs on float rewards/obs; masks bit-exact


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def set_step_draws(env, seed, step, n, n_ranges, obs_width, dev):
    t = lambda a: torch.from_numpy(a).to(dev)
    env.set_draws(**{
        "command:0": t(philox.draws(seed, step, 0, n, n_ranges)),
        "command_reset:0": t(philox.draws(seed, step, 1, n, n_ranges)),
        "episode_length": t(philox.draws(seed, step, 2, n, 1)[:, 0].copy()),
        "obs:policy": t(philox.draws(seed, step, 3, n, obs_width)),
        "spawn": t(philox.draws(seed, step, 4, n, 5)),   # x, y, rot x, rot y, rot z (mdp.reset.randomize_terrain_position)
    })


def _trajectory_env(fix):
    """The package's env a trajectory fixture describes (built); returns (env, n, seed, frame width, history)."""
    from envs import Go2CommandDirectionEnv, Go2RoughTerrainEnv

    n, seed = int(fix["n"]), int(fix["seed"])
    contacts, history = bool(fix["contacts"]), int(fix["history"])
    scene_kwargs = ast.literal_eval(str(fix["scene_kwargs"]))
    variant = str(fix["variant"]) if "variant" in fix else "cmd"
    if variant == "rough":
        env = Go2RoughTerrainEnv(num_envs=n, max_episode_length_s=float(fix["episode_s"]), scene_kwargs=scene_kwargs,
                                 rotation=ast.literal_eval(str(fix["rotation"])), cmd_resample_s=float(fix["cmd_resample_s"]))
    else:
        env = Go2CommandDirectionEnv(num_envs=n, max_episode_length_s=float(fix["episode_s"]), scene_kwargs=scene_kwargs, obs_noise=True,
                                     contacts=contacts, history=history if history > 1 else None, cmd_resample_s=float(fix["cmd_resample_s"]))
    env.build()
    frame = fix["obs"].shape[-1] // history
    return env, n, seed, frame, history


def replay_trajectory(fix, dev="cpu", steps=None):
    """Drive the package's env with a golden fixture's actions/draws; returns per-step outputs like the fixture's."""
    env, n, seed, frame, history = _trajectory_env(fix)
    set_step_draws(env, seed, 0, n, 3, frame, dev)
    obs0, _ = env.reset()
    obs0 = obs0.cpu().numpy().copy()  # the returned tensor is a ring slot, reused two calls later
    out = {k: [] for k in ("obs", "reward", "terminated", "truncated", "command", "episode_length", "max_episode_length", "pos", "quat")}
    logs = []
    T = int(fix["steps"]) if steps is None else steps
    for t in range(T):
        set_step_draws(env, seed, t + 1, n, 3, frame, dev)
        obs, rew, term, trunc, extras = env.step(torch.from_numpy(fix["actions"][t]).to(dev))
        out["obs"].append(obs.cpu().numpy().copy())
        out["reward"].append(rew.cpu().numpy().copy())
        out["terminated"].append(term.cpu().numpy().copy())
        out["truncated"].append(trunc.cpu().numpy().copy())
        out["command"].append(env.velocity_command._command.cpu().numpy().copy())
        out["episode_length"].append(env.episode_length.cpu().numpy().copy())
        out["max_episode_length"].append(env.max_episode_length.cpu().numpy().copy())
        out["pos"].append(env.robot.get_pos().cpu().numpy().copy())
        out["quat"].append(env.robot.get_quat().cpu().numpy().copy())
        logs.append({k: float(v) for k, v in extras["episode"].items()})
    res = {k: np.stack(v) for k, v in out.items()}
    res["obs0"] = obs0
    res["logs"] = logs
    return res


def compare_trajectory(fix, res, T=None, tol=FLOAT_TOL):
    T = len(res["reward"]) if T is None else T
    np.testing.assert_allclose(res["obs0"], fix["obs0"], atol=tol, rtol=0)
    for t in range(T):
        assert np.array_equal(res["terminated"][t], fix["terminated"][t]), f"terminated mask differs at step {t}"
        assert np.array_equal(res["truncated"][t], fix["truncated"][t]), f"truncated mask differs at step {t}"
        assert np.array_equal(res["episode_length"][t], fix["episode_length"][t]), f"episode_length differs at step {t}"
        assert np.array_equal(res["max_episode_length"][t], fix["max_episode_length"][t]), f"max_episode_length differs at step {t}"
        np.testing.assert_allclose(res["command"][t], fix["command"][t], atol=tol, rtol=0, err_msg=f"command step {t}")
        np.testing.assert_allclose(res["reward"][t], fix["reward"][t], atol=tol, rtol=0, err_msg=f"reward step {t}")
        np.testing.assert_allclose(res["obs"][t], fix["obs"][t], atol=tol, rtol=0, err_msg=f"obs step {t}")
        if "pos" in fix:  # base pose after the step's reset (terrain spawn: random x / y, terrain height, random yaw)
            np.testing.assert_allclose(res["pos"][t], fix["pos"][t], atol=tol, rtol=0, err_msg=f"base pos step {t}")
            np.testing.assert_allclose(res["quat"][t], fix["quat"][t], atol=tol, rtol=0, err_msg=f"base quat step {t}")
    keys = [str(k) for k in fix["log_keys"]]
    for t in range(T):
        want = {k: fix["log_values"][t, j] for j, k in enumerate(keys) if not np.isnan(fix["log_values"][t, j])}
        got = res["logs"][t]
        assert set(got) == set(want), f"log keys differ at step {t}: {sorted(got)} vs {sorted(want)}"
        for k in want:
            assert abs(got[k] - want[k]) <= 1e-5 + 1e-5 * abs(want[k]), f"log {k} at step {t}: {got[k]} vs {want[k]}"


# ----------------------------------------------------------------------------------------------------
# The reference's shipped task configs (tests/example_cases.py, fixtures traj_ex_<name>.npz)
# ----------------------------------------------------------------------------------------------------
GAIT_FIELDS = ("foot_offset", "foot_height", "gait_period", "gait_time", "gait_phase", "clock_input", "gait_selected")


def _command_managers(env):
    return list(env.managers["command"])


def set_example_draws(env, seed, step, dev):
    """Same draw convention as tools/gen_golden.py: kind 0/1 command step/reset, 2 episode length, 3 observation noise,
    4 terrain spawn, 5/6 gait step/reset ([N,3]: gait select, foot clearance, gait period)."""
    n = env.num_envs
    t = lambda a: torch.from_numpy(a).to(dev)
    d = {"episode_length": t(philox.draws(seed, step, 2, n, 1)[:, 0].copy()), "spawn": t(philox.draws(seed, step, 4, n, 5))}
    for i, m in enumerate(_command_managers(env)):
        r = m._command.shape[1] if hasattr(m, "_command") else 0
        if getattr(m, "_gf_native_gait", False):
            d[f"gait:{i}"] = t(philox.draws(seed, step, 5, n, 3))
            d[f"gait_reset:{i}"] = t(philox.draws(seed, step, 6, n, 3))
        elif r > 0:
            d[f"command:{i}"] = t(philox.draws(seed, step, 0, n, r))
            d[f"command_reset:{i}"] = t(philox.draws(seed, step, 1, n, r))
    env.set_draws(**d)


def replay_example(fix, case, env, dev="cpu", user_gait_cls=None, compact=False):
    """Drive ``env`` (a task config of tests/example_cases.py, built here) with the fixture's actions and draws.  ``compact``: keep per
    step only what a compact fixture holds (compact_example's sums and sample, computed on the device) — a full trajectory at a
    timed size is hundreds of MB of host copies."""
    import example_cases

    seed, n = int(fix["seed"]), case["n"]
    cur = {"step": 0}
    if user_gait_cls is not None:
        import gait_rng
        gait_rng.install(user_gait_cls, lambda mgr, phase: philox.draws(seed, cur["step"], 5 if phase == "step" else 6, n, 3))
    env.build()
    for attr, sec in case["resample"].items():
        getattr(env, attr).resample_time_sec = sec
    set_example_draws(env, seed, 0, dev)
    obs0, _ = env.reset()
    f = lambda t: t.detach().cpu().numpy().copy()
    out = {"obs0": f(obs0)}
    sample_idx = torch.from_numpy(at_size_sample(n)).to(dev) if compact else None
    rec = {}
    logs = []
    for t in range(case["steps"]):
        if t in case["events"]:
            example_cases.apply_event(env, case["events"][t])
        cur["step"] = t + 1
        set_example_draws(env, seed, t + 1, dev)
        obs, rew, term, trunc, extras = env.step(torch.from_numpy(fix["actions"][t]).to(dev))
        vals = dict(obs=obs, reward=rew, terminated=term, truncated=trunc, episode_length=env.episode_length,
                    max_episode_length=env.max_episode_length, pos=env.robot.get_pos(), quat=env.robot.get_quat())
        for k, m in enumerate(_command_managers(env)):
            vals[f"command{k}"] = m.command
        for name, o in extras["observations"].items():
            if name != "policy":
                vals["obs_" + name] = o
        g = getattr(env, "gait_command_manager", None)
        if g is not None:
            for fld in GAIT_FIELDS:
                vals["gait_" + fld] = getattr(g, fld if fld != "gait_selected" else "_gait_selected")
        if compact:
            for k, v in vals.items():
                a = v.reshape(n, -1)
                if k in _EX_EXACT:
                    rec.setdefault(k + "_sum", []).append(int(a.long().sum()))
                else:
                    d = a.double()
                    rec.setdefault(k + "_sum", []).append(float(d.sum()))
                    rec.setdefault(k + "_sumsq", []).append(float((d * d).sum()))
                rec.setdefault(k + "_sample", []).append(f(a[sample_idx]))
        else:
            for k, v in vals.items():
                rec.setdefault(k, []).append(f(v))
        logs.append({k: float(v) for k, v in extras["episode"].items()})
    if compact:
        out = {"n": np.int64(n), "obs0_sample": out["obs0"][at_size_sample(n)], "obs0_sum": np.asarray(out["obs0"], dtype=np.float64).sum()}
        out.update({k: (np.stack(v) if k.endswith("_sample") else np.asarray(v)) for k, v in rec.items()})
    else:
        out.update({k: np.stack(v) for k, v in rec.items()})
    out["logs"] = logs
    return out


def compare_example(fix, res, tol=FLOAT_TOL):
    T = len(res["reward"])
    np.testing.assert_allclose(res["obs0"], fix["obs0"], atol=tol, rtol=0)
    exact = ("terminated", "truncated", "episode_length", "max_episode_length", "gait_gait_selected")
    skip = ("actions", "obs0", "log_keys", "log_values", "seed", "example")
    keys = [k for k in fix.files if k not in skip]
    for k in keys:
        assert k in res, f"fixture field {k} not produced"
    for t in range(T):
        for k in keys:
            if k in exact:
                assert np.array_equal(res[k][t], fix[k][t]), f"{k} differs at step {t}"
            else:
                np.testing.assert_allclose(res[k][t].reshape(fix[k][t].shape), fix[k][t], atol=tol, rtol=0, err_msg=f"{k} step {t}")
    lkeys = [str(k) for k in fix["log_keys"]]
    for t in range(T):
        want = {k: fix["log_values"][t, j] for j, k in enumerate(lkeys) if not np.isnan(fix["log_values"][t, j])}
        got = res["logs"][t]
        assert set(got) == set(want), f"log keys differ at step {t}: {sorted(got)} vs {sorted(want)}"
        for k in want:
            assert abs(got[k] - want[k]) <= 1e-5 + 1e-5 * abs(want[k]), f"log {k} at step {t}: {got[k]} vs {want[k]}"


_EX_SKIP = ("actions", "obs0", "log_keys", "log_values", "seed", "example", "logs")
_EX_EXACT = ("terminated", "truncated", "episode_length", "max_episode_length", "gait_gait_selected")


def example_actions(case):
    """The action stream of tools/gen_golden.py run_example (a compact fixture does not store it)."""
    rng = np.random.RandomState(7)
    acts = []
    for t in range(case["steps"]):
        act = rng.standard_normal((case["n"], case["dofs"])).astype(np.float32)
        if t == 5:
            act[0, 0] = 1e9
        acts.append(act)
    return np.stack(acts)


def compact_example(out, n):
    """An example trajectory (every recorded field, [steps, n, …]) reduced to what travels: per step the sum over all envs (f64, or
    int64 for masks / counters / indices), for floats the sum of squares, and every value of a strided 64-env sample."""
    idx = at_size_sample(n)
    c = {"n": np.int64(n), "compact": np.int64(1), "sample": idx, "obs0_sample": out["obs0"][idx], "obs0_sum": np.asarray(out["obs0"], dtype=np.float64).sum()}
    for k, v in out.items():
        if k in _EX_SKIP or not isinstance(v, np.ndarray):
            continue
        a = v.reshape(v.shape[0], n, -1)
        if k in _EX_EXACT:
            c[k + "_sum"] = a.astype(np.int64).sum(axis=(1, 2))
        else:
            d = a.astype(np.float64)
            c[k + "_sum"], c[k + "_sumsq"] = d.sum(axis=(1, 2)), (d * d).sum(axis=(1, 2))
        c[k + "_sample"] = a[:, idx]
    for k in ("log_keys", "log_values", "seed", "example"):
        if k in out:
            c[k] = out[k]
    return c


def compare_compact_example(fix, got, logs, tol=FLOAT_TOL):
    n = int(fix["n"])
    np.testing.assert_allclose(got["obs0_sample"], fix["obs0_sample"], atol=tol, rtol=0)
    fields = sorted(k[:-7] for k in fix.files if k.endswith("_sample") and k != "obs0_sample")
    assert fields and all(f + "_sample" in got for f in fields), [f for f in fields if f + "_sample" not in got]
    for k in fields:
        if k in _EX_EXACT:
            assert np.array_equal(got[k + "_sum"], fix[k + "_sum"]), f"{k}: per-step integer sum over all envs differs"
            assert np.array_equal(got[k + "_sample"], fix[k + "_sample"]), f"{k}: sampled envs differ"
        else:
            np.testing.assert_allclose(got[k + "_sample"], fix[k + "_sample"], atol=tol, rtol=0, err_msg=f"{k}: sampled envs")
            width = fix[k + "_sample"].shape[-1]
            np.testing.assert_allclose(got[k + "_sum"], fix[k + "_sum"], atol=n * width * tol, rtol=1e-13, err_msg=f"{k}: sum over all envs")
            scale = 2.0 * float(np.abs(fix[k + "_sample"]).max() + 1.0) * 4.0
            np.testing.assert_allclose(got[k + "_sumsq"], fix[k + "_sumsq"], atol=n * width * tol * scale, rtol=1e-13, err_msg=f"{k}: sum of squares")
    lkeys = [str(k) for k in fix["log_keys"]]
    for t in range(len(logs)):
        want = {k: fix["log_values"][t, j] for j, k in enumerate(lkeys) if not np.isnan(fix["log_values"][t, j])}
        assert set(logs[t]) == set(want), f"log keys differ at step {t}: {sorted(logs[t])} vs {sorted(want)}"
        for k in want:
            assert abs(logs[t][k] - want[k]) <= 1e-5 + 1e-5 * abs(want[k]), f"log {k} at step {t}: {logs[t][k]} vs {want[k]}"


# ----------------------------------------------------------------------------------------------------
# The reference at the sizes that get timed (tools/gen_golden.py at_size; fixtures atsize_go2_<n>.npz)
# ----------------------------------------------------------------------------------------------------
AT_SIZE_SAMPLE = 64   # envs of the strided sample a compact fixture keeps


def at_size_sample(n):
    return np.arange(AT_SIZE_SAMPLE, dtype=np.int64) * (n // AT_SIZE_SAMPLE) + (n // AT_SIZE_SAMPLE) // 2


def at_size_actions(n, steps):
    """The action stream of tools/gen_golden.py run_trajectory (not stored in the compact fixtures: 63 MB at 65 536 envs)."""
    rng = np.random.RandomState(7)
    acts = []
    for t in range(steps):
        act = rng.standard_normal((n, 12)).astype(np.float32)
        if t == 5:
            act[0, 0] = 1e9  # clip path
        acts.append(act)
    return np.stack(acts)


def compact_at_size(out, n):
    """What of an at-size trajectory travels to the GPU box: per-step mask popcounts, sums and sums of squares of reward,
    observation and command (f64), integer sums of the episode counters, the per-step logs, and every output for a strided sample
    of 64 envs.  Applied to the reference's trajectory by the generator and to the package's by the tests."""
    idx = at_size_sample(n)
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    c = {"n": np.int64(n), "sample": idx, "obs0_sample": out["obs0"][idx], "obs0_sum": f64(out["obs0"]).sum()}
    for k in ("obs", "reward", "command"):
        a = f64(out[k]).reshape(out[k].shape[0], -1)
        c[k + "_sum"], c[k + "_sumsq"] = a.sum(axis=1), (a * a).sum(axis=1)
        c[k + "_sample"] = out[k][:, idx]
    for k in ("terminated", "truncated"):
        c[k + "_count"] = out[k].reshape(out[k].shape[0], -1).sum(axis=1).astype(np.int64)
        c[k + "_sample"] = out[k][:, idx]
    for k in ("episode_length", "max_episode_length"):
        c[k + "_sum"] = out[k].astype(np.int64).sum(axis=1)
        c[k + "_sample"] = out[k][:, idx]
    for k in ("log_keys", "log_values", "seed", "steps", "contacts", "history", "episode_s", "cmd_resample_s", "scene_kwargs", "variant", "rotation"):
        if k in out:
            c[k] = out[k]
    return c


def replay_at_size(fix, dev="cpu"):
    """Run the package on a compact at-size fixture's inputs (regenerated actions, Philox draws) and compact its outputs the same
    way — streaming: per step only the sums and the sampled rows are kept (the full trajectory at 65 536 envs is 0.5 GB)."""
    n, steps = int(fix["n"]), int(fix["steps"])
    full = {k: fix[k] for k in ("n", "seed", "contacts", "history", "scene_kwargs", "variant", "episode_s", "cmd_resample_s", "steps", "rotation")}
    full["obs"] = np.empty((0, fix["obs_sample"].shape[-1]), dtype=np.float32)   # (the frame width is read off it)
    env, n, seed, frame, history = _trajectory_env(full)
    idx = torch.from_numpy(at_size_sample(n)).to(dev)
    set_step_draws(env, seed, 0, n, 3, frame, dev)
    obs0, _ = env.reset()
    c = {"n": np.int64(n), "obs0_sample": obs0[idx].cpu().numpy().copy(), "obs0_sum": float(obs0.double().sum())}
    rec: dict = {}
    logs = []
    rng = np.random.RandomState(7)   # at_size_actions(), one step at a time
    for t in range(steps):
        act = rng.standard_normal((n, 12)).astype(np.float32)
        if t == 5:
            act[0, 0] = 1e9
        set_step_draws(env, seed, t + 1, n, 3, frame, dev)
        obs, rew, term, trunc, extras = env.step(torch.from_numpy(act).to(dev))
        vals = {"obs": obs, "reward": rew, "command": env.velocity_command._command}
        for k, v in vals.items():
            d = v.double()
            rec.setdefault(k + "_sum", []).append(float(d.sum()))
            rec.setdefault(k + "_sumsq", []).append(float((d * d).sum()))
            rec.setdefault(k + "_sample", []).append(v[idx].cpu().numpy().copy())
        for k, v in (("terminated", term), ("truncated", trunc)):
            rec.setdefault(k + "_count", []).append(int(v.sum()))
            rec.setdefault(k + "_sample", []).append(v[idx].cpu().numpy().copy())
        for k, v in (("episode_length", env.episode_length), ("max_episode_length", env.max_episode_length)):
            rec.setdefault(k + "_sum", []).append(int(v.long().sum()))
            rec.setdefault(k + "_sample", []).append(v[idx].cpu().numpy().copy())
        logs.append({k: float(v) for k, v in extras["episode"].items()})
    for k, v in rec.items():
        c[k] = np.stack(v) if k.endswith("_sample") else np.asarray(v)
    return c, logs


def compare_at_size(fix, got, logs, tol=FLOAT_TOL):
    n, T = int(fix["n"]), int(fix["steps"])
    np.testing.assert_allclose(got["obs0_sample"], fix["obs0_sample"], atol=tol, rtol=0)
    for k in ("terminated", "truncated"):
        assert np.array_equal(got[k + "_count"], fix[k + "_count"]), f"{k}: popcount per step differs: {got[k + '_count']} vs {fix[k + '_count']}"
        assert np.array_equal(got[k + "_sample"], fix[k + "_sample"]), f"{k}: sampled envs differ"
    for k in ("episode_length", "max_episode_length"):
        assert np.array_equal(got[k + "_sum"], fix[k + "_sum"]), f"{k}: per-step integer sum differs"
        assert np.array_equal(got[k + "_sample"], fix[k + "_sample"]), f"{k}: sampled envs differ"
    for k in ("obs", "reward", "command"):
        np.testing.assert_allclose(got[k + "_sample"], fix[k + "_sample"], atol=tol, rtol=0, err_msg=f"{k}: sampled envs")
        width = fix[k + "_sample"].reshape(T, AT_SIZE_SAMPLE, -1).shape[-1]
        # N*width values per step, each within tol of the reference's: the f64 sums agree to N*width*tol, the sums of squares to
        # 2*max|x|*tol per value (|x| <= 1e9 only for the one clipped action at step 5, which no output carries)
        # (rtol: the one env whose clipped action makes its reward ~ -2e14 at step 5 — there the f64 sum's own last bit is 2^-5)
        np.testing.assert_allclose(got[k + "_sum"], fix[k + "_sum"], atol=n * width * tol, rtol=1e-13, err_msg=f"{k}: sum over all envs")
        scale = 2.0 * float(np.abs(fix[k + "_sample"]).max() + 1.0) * 4.0
        np.testing.assert_allclose(got[k + "_sumsq"], fix[k + "_sumsq"], atol=n * width * tol * scale, rtol=1e-13, err_msg=f"{k}: sum of squares over all envs")
    keys = [str(k) for k in fix["log_keys"]]
    for t in range(T):
        want = {k: fix["log_values"][t, j] for j, k in enumerate(keys) if not np.isnan(fix["log_values"][t, j])}
        assert set(logs[t]) == set(want), f"log keys differ at step {t}"
        for k in want:
            assert abs(logs[t][k] - want[k]) <= 1e-5 + 1e-5 * abs(want[k]), f"log {k} at step {t}: {logs[t][k]} vs {want[k]}"
