"""
Wrappers — thin mirrors of genesis_forge/wrappers/{wrapper.py, rsl_rl.py, skrl.py}.  O(1) tensor
ops per step; nothing here is on the hot path.  VideoWrapper drives a Genesis camera's
recording calls on the schedule of video.py:13-260 and passes through when the env has no camera (a synthetic scene), so that the
reference's training scripts — all of which wrap their env in it — run unchanged.
"""
from __future__ import annotations

import math
import os
import warnings
from typing import Any, Callable, Optional

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


def capped_cubic_episode_trigger(episode_id: int) -> bool:
    """The default schedule (video.py:13-27): episodes 0, 1, 8, 27, …, k^3, …, 729, then every 1000th."""
    if episode_id < 1000:
        return int(round(episode_id ** (1.0 / 3))) ** 3 == episode_id
    return episode_id % 1000 == 0


class VideoWrapper(Wrapper):
    """Records videos of one env through the Genesis camera at ``env.<camera_attr>`` (ctor and schedule as video.py:90-260): a
    triggered recording of ``video_length_sec`` is saved as ``<out_dir>/<start step>.mp4``; between triggers the running episode is
    recorded in the background so that ``close()`` can save the final one.  The wrapper only makes the camera's own calls
    (``start_recording`` / ``render`` / ``pause_recording`` / ``stop_recording``); rendering is Genesis'.  An env without that camera
    (the synthetic scene) is passed through, with one warning.  With a camera the episode count reads one env's done flags per
    step — a host synchronisation per step, as in the reference.

    State: ``_mode`` is ``None`` (nothing running), ``"active"`` (a triggered clip, saved when its length is reached), ``"background"``
    (the running episode, kept in the camera's buffer) or ``"held"`` (a background clip paused in the buffer, saved by ``close()``)."""

    def __init__(self, env, camera_attr: str = "camera", video_length_sec: int = 8, episode_trigger: Optional[Callable[[int], bool]] = None,
                 step_trigger: Optional[Callable[[int], bool]] = None, out_dir: str = "./videos", fps: int = 60, env_idx: int = 0,
                 filename: Optional[str] = None, record_final_episode: bool = True, logging: bool = True):
        super().__init__(env)
        if episode_trigger is None and step_trigger is None:
            episode_trigger = capped_cubic_episode_trigger
        if (episode_trigger is None) == (step_trigger is None):
            raise AssertionError("Must specify only one trigger")
        self.episode_trigger, self.step_trigger = episode_trigger, step_trigger
        self._where = (camera_attr, out_dir, filename, int(env_idx))
        self._opts = (bool(record_final_episode), bool(logging))
        dt = self.dt
        self._clip_steps = math.ceil(video_length_sec / dt)
        self._frame_every = max(1, round(1.0 / fps / dt))
        self._file_fps = round(1.0 / dt / self._frame_every)
        self._cam = None
        self._mode: Optional[str] = None
        self._steps = self._episodes = 0
        self._clip_from = self._clip_until = 0
        os.makedirs(out_dir, exist_ok=True)

    @property
    def video_length_steps(self) -> int:
        return self._clip_steps

    def build(self) -> None:
        super().build()
        attr = self._where[0]
        self._cam = getattr(self.unwrapped, attr, None)
        if self._cam is None:
            warnings.warn(f"VideoWrapper: {type(self.unwrapped).__name__}.{attr} is not a camera - nothing will be recorded", RuntimeWarning)

    def step(self, actions: torch.Tensor):
        result = self.env.step(actions)
        cam = self._cam
        if cam is None:
            return result
        if self._mode != "active" and self._due():
            self.start_recording("active")
        if self._steps % self._frame_every == 0:
            cam.render()
        if self._mode in ("active", "background") and self._steps >= self._clip_until:
            self.finish_recording()
        k = self._where[3]
        flags = [f for f in (result[2], result[3]) if f is not None]
        if any(bool(f[k]) for f in flags):       # the watched env finished an episode
            self._episodes += 1
            if self._mode not in ("active", "background") and self._opts[0]:
                self.start_recording("background")   # (the last of these is what close() saves as the final episode)
        self._steps += 1
        return result

    def _due(self) -> bool:
        return bool(self.step_trigger(self._steps) if self.episode_trigger is None else self.episode_trigger(self._episodes))

    def start_recording(self, type: str = "active"):
        cam = self._cam
        if cam is None:
            return
        frames = getattr(cam, "_recorded_imgs", None)
        if frames is not None:
            frames.clear()
        self._mode = type
        self._clip_from, self._clip_until = self._steps, self._steps + self._clip_steps
        cam.start_recording()

    def finish_recording(self):
        cam, mode = self._cam, self._mode
        if cam is None or mode is None:
            return
        if mode == "background":        # keep it in the camera's buffer: only close() turns it into a file
            cam.pause_recording()
            self._mode = "held"
            return
        _attr, out_dir, filename, _k = self._where
        target = os.path.join(out_dir, filename or f"{self._clip_from}.mp4")
        if self._opts[1]:
            print(f"Saving recording to {target}")
        cam.stop_recording(target, fps=self._file_fps)
        self._mode, self._clip_until = None, 0

    def close(self):
        if self._cam is not None and self._mode is not None:
            if self._mode == "background":
                self._mode = "held"
            self.finish_recording()
        return self.env.close()
