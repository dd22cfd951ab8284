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
    """``dones = terminated | truncated``; observations and time-outs in extras (rsl_rl.py:11-119)."""

    can_be_wrapped = False

    def step(self, actions: torch.Tensor):
        obs, rewards, terminated, truncated, extras = self.env.step(actions)
        dones = terminated | truncated
        if extras is None:
            extras = {}
        extras.setdefault("observations", {})
        if "critic" not in extras["observations"]:
            extras["observations"]["critic"] = obs
        extras["time_outs"] = truncated
        return obs, rewards, dones, extras

    def reset(self):
        obs, extras = self.env.reset()
        return obs, extras

    def get_observations(self):
        obs = self.env.get_observations()
        return obs, self.env.extras


class SkrlEnvWapper(Wrapper):
    """skrl expects ``[N, 1]`` shaped rewards / dones (skrl.py:36-54)."""

    can_be_wrapped = False

    def step(self, actions: torch.Tensor):
        obs, rewards, terminated, truncated, extras = self.env.step(actions)
        return obs, rewards.unsqueeze(-1), terminated.unsqueeze(-1), truncated.unsqueeze(-1), extras

    def reset(self):
        obs, extras = self.env.reset()
        return obs, extras
