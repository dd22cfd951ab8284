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


def replay_trajectory(fix, dev="cpu", steps=None):
    """Drive the package's env with a golden fixture's actions/draws; returns per-step outputs like the fixture's."""
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
