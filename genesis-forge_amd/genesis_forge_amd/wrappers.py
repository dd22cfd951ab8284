"""
Wrappers — thin mirrors of genesis_forge/wrappers/{wrapper.py, rsl_rl.py, skrl.py}.  O(1) tensor
ops per step; nothing here is on the hot path.  VideoWrapper (camera I/O) is
out of scope.
"""
from __future__ import annotations

from typing import Any

import torch


class Wrapper:
    """Pass-through base wrapper (wrapper.py:11-121)."""

    can_be_wrapped = True

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def build(self):
        return self.env.build()

    def step(self, actions):
        return self.env.step(actions)

    def reset(self, *a, **k):
        return self.env.reset(*a, **k)

    def get_observations(self):
        return self.env.get_observations()

    def close(self):
        return self.env.close()


class RslRlWrapper(Wrapper):
    """rsl_rl's env interface (rsl_rl.py:11-135): ``dones = terminated | truncated``, the policy observation as the critic's
    when the env has none, and — with rsl-rl-lib 3.0 or later installed — observations as a ``TensorDict`` of the env's
    observation groups (``extras["observations"]``), which is what that version's runners index by group name."""

    can_be_wrapped = False

    def __init__(self, env):
        super().__init__(env)
        self.rsl3 = False
        try:
            from importlib import metadata
            self.rsl3 = int(metadata.version("rsl-rl-lib").split(".")[0]) >= 3
        except Exception:
            pass

    @property
    def device(self):
        from . import gs
        return gs.device

    def step(self, actions: torch.Tensor):
        obs, rewards, terminated, truncated, extras = self.env.step(actions)
        dones = terminated | truncated
        # (time-outs: extras["time_outs"] is the termination manager's, termination_manager.py:189 — kept here for envs without one)
        extras = self._add_observations_to_extras(obs, extras)
        extras.setdefault("time_outs", truncated)
        return self._format_obs_group(obs, extras), rewards, dones, extras

    def reset(self):
        obs, extras = self.env.reset()
        return self._format_obs_group(obs, extras), extras

    def get_observations(self):
        obs = self.env.get_observations()
        if self.rsl3:   # rsl_rl 3.0+ wants the observations only (rsl_rl.py:78-81)
            return self._format_obs_group(obs, self.env.extras)
        return obs, self._add_observations_to_extras(obs, self.env.extras)

    @staticmethod
    def _add_observations_to_extras(obs, extras):
        if extras is None:
            extras = {}
        if "observations" not in extras:
            extras["observations"] = {}
        if "critic" not in extras["observations"]:
            extras["observations"]["critic"] = obs
        return extras

    def _format_obs_group(self, obs, extras):
        """rsl_rl 3.0+: the observation groups as a TensorDict (rsl_rl.py:101-119); earlier versions: the policy tensor."""
        if not self.rsl3:
            return obs
        from tensordict import TensorDict
        from . import gs
        if extras is not None and "observations" in extras:
            groups = extras["observations"]
            return groups if isinstance(groups, TensorDict) else TensorDict(dict(groups), device=gs.device)
        return TensorDict({"policy": obs}, batch_size=[obs.shape[0]], device=gs.device)


class SkrlEnvWapper(Wrapper):
    """skrl's env interface (skrl.py:9-84; the reference also derives from skrl's own Wrapper base, which only stores the env):
    ``[N, 1]`` shaped rewards / terminations / time-outs."""

    can_be_wrapped = False

    def __init__(self, env):
        super().__init__(env)
        self._env = env   # (the attribute name skrl's base class uses)

    @property
    def action_space(self):
        return self.env.action_space

    @property
    def observation_space(self):
        return self.env.observation_space

    def step(self, actions: torch.Tensor):
        obs, rewards, terminated, truncated, extras = self.env.step(actions)
        return obs, rewards.unsqueeze(1), terminated.unsqueeze(1), truncated.unsqueeze(1), extras

    def reset(self):
        return self.env.reset()

    def state(self):
        return self.env.state()

    def render(self, *args, **kwargs):
        """Not implemented for these environments (skrl.py:69-73)."""
        return None
