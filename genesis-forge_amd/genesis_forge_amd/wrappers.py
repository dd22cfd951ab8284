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
    triggered ("active") recording of ``video_length_sec`` is saved as ``<out_dir>/<start step>.mp4``; between triggers a
    "background" recording of the running episode is kept so that ``close()`` can save the final one.  The wrapper only makes the
    camera's own calls (``start_recording`` / ``render`` / ``pause_recording`` / ``stop_recording``); rendering is Genesis'.
    An env without that camera (the synthetic scene) is passed through, with one warning.  With a camera the episode count reads
    one env's done flags per step — a host synchronisation per step, as in the reference."""

    def __init__(self, env, camera_attr: str = "camera", video_length_sec: int = 8, episode_trigger: Optional[Callable[[int], bool]] = None,
                 step_trigger: Optional[Callable[[int], bool]] = None, out_dir: str = "./videos", fps: int = 60, env_idx: int = 0,
                 filename: Optional[str] = None, record_final_episode: bool = True, logging: bool = True):
        super().__init__(env)
        self._is_recording = False
        self._has_recording_buffer = False
        self._recording_type: Optional[str] = "background"
        self._logging = logging
        self._current_step = self._current_episode = 0
        self._recording_start_step = self._recording_stop_step = 0
        self._record_final_episode = record_final_episode
        self._cam = None
        self._camera_attr, self._out_dir, self._filename, self._env_idx = camera_attr, out_dir, filename, env_idx
        self._video_length_steps = math.ceil(video_length_sec / self.dt)
        self._steps_per_frame = max(1, round(1.0 / fps / self.dt))
        self._actual_fps = round(1.0 / self.dt / self._steps_per_frame)
        if episode_trigger is None and step_trigger is None:
            episode_trigger = capped_cubic_episode_trigger
        assert (episode_trigger is None) != (step_trigger is None), "Must specify only one trigger"
        self.episode_trigger, self.step_trigger = episode_trigger, step_trigger
        os.makedirs(self._out_dir, exist_ok=True)

    @property
    def video_length_steps(self) -> int:
        return self._video_length_steps

    def build(self) -> None:
        super().build()
        self._cam = getattr(self.unwrapped, self._camera_attr, None)
        if self._cam is None:
            warnings.warn(f"VideoWrapper: {type(self.unwrapped).__name__}.{self._camera_attr} is not a camera - nothing will be recorded", RuntimeWarning)

    def step(self, actions: torch.Tensor):
        out = self.env.step(actions)
        if self._cam is None:
            return out
        _obs, _rew, terminateds, truncateds, _extras = out
        self._check_recording_trigger()
        if self._current_step % self._steps_per_frame == 0:
            self._cam.render()
        if self._is_recording and self._recording_stop_step <= self._current_step:
            self.finish_recording()
        done = bool(terminateds is not None and terminateds[self._env_idx]) or bool(truncateds is not None and truncateds[self._env_idx])
        if done:
            self._current_episode += 1
            if not self._is_recording and self._record_final_episode:
                self.start_recording("background")   # (the last of these is what close() saves as the final episode)
        self._current_step += 1
        return out

    def close(self):
        if self._cam is not None and (self._is_recording or self._has_recording_buffer):
            self.finish_recording()
        return self.env.close()

    def start_recording(self, type: str = "active"):
        if self._cam is None:
            return
        if hasattr(self._cam, "_recorded_imgs"):
            self._cam._recorded_imgs.clear()
        self._is_recording, self._has_recording_buffer, self._recording_type = True, False, type
        self._recording_start_step = self._current_step
        self._recording_stop_step = self._current_step + self._video_length_steps
        self._cam.start_recording()

    def finish_recording(self):
        if self._cam is None or (not self._is_recording and not self._has_recording_buffer):
            return
        if self._recording_type == "active" or (not self._is_recording and self._has_recording_buffer):
            path = os.path.join(self._out_dir, self._filename or f"{self._recording_start_step}.mp4")
            if self._logging:
                print(f"Saving recording to {path}")
            self._cam.stop_recording(path, fps=self._actual_fps)
            self._has_recording_buffer = False
        else:
            self._cam.pause_recording()
            self._has_recording_buffer = True
        self._is_recording, self._recording_type, self._recording_stop_step = False, None, 0

    def _check_recording_trigger(self) -> bool:
        if self._is_recording and self._recording_type == "active":
            return False
        record = bool(self.episode_trigger(self._current_episode) if self.episode_trigger is not None else self.step_trigger(self._current_step))
        if record:
            self.start_recording()
        return record
