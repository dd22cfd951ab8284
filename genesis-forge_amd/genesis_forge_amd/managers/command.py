"""
Command managers — API mirror of genesis_forge/managers/command/{command_manager.py, velocity_command.py}.

``step`` / ``reset`` / ``resample_command`` (command_manager.py:152-170, 290-303) run as
``gf_command_step`` (Phase B5): the resample predicate ``episode_length % resample_steps == 0`` is
evaluated per env on device and the draws come from Philox (or from parity-mode draws), so the
reference's ``nonzero()`` host sync and its per-range scatter launches disappear.  Ranges are read
from ``self._range`` on every call, so curricula that mutate ``range`` keep working (:293-298).

Out of scope (viewer / human-input only): debug arrows, gamepad HID reading.
``use_external_controller`` is kept because it is plain tensor plumbing.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .. import _native as nat
from .. import gs
from .action import _tag
from .base import BaseManager

CommandRange = tuple | dict


class CommandManager(BaseManager):
    """Generates a command from uniform distribution(s) (ctor as command_manager.py:60-81)."""

    _fused_reset = True

    def __init__(self, env, range: CommandRange, resample_time_sec: float = 5.0):
        super().__init__(env, type="command")
        self._range = range
        self.resample_time_sec = resample_time_sec
        self._external_controller = None
        self._gamepad_cfg = None
        self._gamepad_axis_command_buffer = None
        num_ranges = len(range) if isinstance(range, dict) else 1
        if num_ranges > nat.GF_MAX_RANGES:
            raise ValueError(f"CommandManager supports at most {nat.GF_MAX_RANGES} ranges")
        self._command = torch.zeros(env.num_envs, num_ranges, device=gs.device)
        self._range_idx = {}
        if isinstance(range, dict):
            self._range_idx = {key: i for i, key in enumerate(range.keys())}
        self._args_by_mode = {m: nat.GfCommandArgs() for m in (nat.GF_CMD_STEP, nat.GF_CMD_MASKED, nat.GF_CMD_ALL)}
        self._index = len(env.managers["command"]) - 1 if hasattr(env, "managers") else 0

    # -- properties (command_manager.py:87-130) -------------------------------------------------------
    @property
    def command(self) -> torch.Tensor:
        if self._external_controller is not None:
            return self._external_controller(self.env.step_count)
        return _tag(self._command, ("cmd", self))

    @property
    def range(self) -> CommandRange:
        return self._range

    @range.setter
    def range(self, range: CommandRange):
        num = len(range) if isinstance(range, dict) else 1
        if num != self._command.shape[1]:
            raise ValueError(f"Cannot change the shape of the CommandManager range. Expected size: {self._command.shape[1]}, got {num}")
        if type(range) != type(self._range):
            raise ValueError(f"Cannot change the base type of the CommandManager range. Expected type: {type(self._range)}, got {type(range)}")
        if isinstance(range, dict) and set(range.keys()) != set(self._range.keys()):
            raise ValueError(f"Cannot change the dict keys of the CommandManager range. Expected keys: {set(self._range.keys())}, got {set(range.keys())}")
        self._range = range

    @property
    def resample_time_sec(self) -> float:
        return self._resample_time_sec

    @resample_time_sec.setter
    def resample_time_sec(self, resample_time_sec: float):
        self._resample_time_sec = resample_time_sec
        self._resample_steps = int(resample_time_sec / self.env.dt)

    # -- operations -----------------------------------------------------------------------------------
    def get_command(self, key: str) -> torch.Tensor:
        if not isinstance(self._range, dict):
            raise ValueError("The range is not a dict")
        return self._command[:, self._range_idx[key]]

    def get_command_idx(self, key: str) -> int:
        if not isinstance(self._range, dict):
            raise ValueError("The range is not a dict")
        return self._range_idx[key]

    def _ranges(self) -> list:
        return list(self._range.values()) if isinstance(self._range, dict) else [self._range]

    def _launch(self, mode: int, mask=None, mask2=None, draws_key: Optional[str] = None) -> None:
        env = self.env
        a = self._args_by_mode[mode]
        a.num_envs, a.num_ranges, a.mode = env.num_envs, self._command.shape[1], mode
        a.resample_steps = self._resample_steps
        a.episode_length = env.episode_length.data_ptr()
        a.mask = None if mask is None else mask.data_ptr()
        a.mask2 = None if mask2 is None else mask2.data_ptr()
        draws = env.take_draws(draws_key) if draws_key else None
        self._keep = (draws, mask, mask2)
        a.draws = None if draws is None else draws.data_ptr()
        a.seed, a.stream, a.env_offset = env._rng_seed, env.next_stream(), env.env_offset
        for i, r in enumerate(self._ranges()):  # re-read every call: curricula change ranges (:293-298)
            a.lo[i], a.hi[i] = float(r[0]), float(r[1])
        a.command = self._command.data_ptr()
        a.stats = env.stats.ptr if mode == nat.GF_CMD_STEP else None
        env.backend.call("command_step", a, owner=self)

    def _user_resamples(self) -> bool:
        """A user subclass supplies its own ``resample_command`` (e.g. examples/gait_trainer/gait_command_manager.py:185-211):
        it must be called with env ids exactly as the reference's base class calls it (command_manager.py:152-170)."""
        for klass in type(self).__mro__:
            if "resample_command" in klass.__dict__:
                return not klass.__module__.startswith(__package__.rsplit(".", 1)[0] + ".")
        return False

    def _can_fuse_reset(self) -> bool:
        return not self._user_resamples()

    def step(self):
        """Resample where ``episode_length % resample_steps == 0`` (command_manager.py:152-162)."""
        if not self.enabled or self._external_controller is not None:
            return
        if self._user_resamples():
            ids = (self.env.episode_length % self._resample_steps == 0).nonzero(as_tuple=False).reshape(-1)
            self.resample_command(ids)
            return
        self._launch(nat.GF_CMD_STEP, draws_key=f"command:{self._index}")

    def reset(self, env_ids: list[int] | None = None):
        """command_manager.py:164-170"""
        if not self.enabled:
            return
        self.resample_command(env_ids)

    def resample_command(self, env_ids):
        """New command for the given env ids (command_manager.py:290-303).  ``None`` → all envs."""
        if env_ids is None:
            self._launch(nat.GF_CMD_ALL, draws_key=f"command_reset:{self._index}")
        else:
            self._launch(nat.GF_CMD_MASKED, mask=self.env._ids_to_mask(env_ids), draws_key=f"command_reset:{self._index}")

    def _after_fused_reset(self, mask, mask2) -> None:
        if self.enabled:
            self._launch(nat.GF_CMD_MASKED, mask=mask, mask2=mask2, draws_key=f"command_reset:{self._index}")

    def observation(self, env) -> torch.Tensor:
        return self.command

    def use_external_controller(self, controller: Callable[[int], torch.Tensor]):
        """Bypass the internal generator with ``controller(step_count) -> [N, R]`` (command_manager.py:176-207)."""
        self._external_controller = controller
        self.env.invalidate_trace()

    def _trace_patch(self, args):
        """Per-step refresh of a recorded command launch: ranges are re-read (curricula mutate them,
        command_manager.py:293-298) and the Philox stream id advances exactly as in the unrecorded path."""
        env = self.env

        seen = [None]

        def patch(_actions, a=args, env=env, self=self, seen=seen):
            # ranges and the resample period are live (curricula edit the range lists in place): compare cheaply, rewrite the
            # descriptor only when something changed
            rng = self._range
            key = (self._resample_steps, *[v for r in (rng.values() if type(rng) is dict else (rng,)) for v in r])
            if key != seen[0]:
                seen[0] = key
                for i, r in enumerate(self._ranges()):
                    a.lo[i], a.hi[i] = float(r[0]), float(r[1])
                a.resample_steps = self._resample_steps   # `resample_time_sec` is a live property (command_manager.py:121-130)

        return patch

    def _trace_native(self, args) -> list:
        """The launch's Philox stream as a native patch (what resample draws with env.next_stream())."""
        return [nat.GfReplayPatch(nat.GF_PATCH_STREAM, 0, nat.field_addr(args, "stream"), None, None)]

    def use_gamepad(self, gamepad, range_axis):
        """Map gamepad axes onto the command ranges (command_manager.py:209-288).  The HID reader itself is
        out of scope; any object with ``state.axis(i) -> float in [-1, 1]`` works."""
        self._external_controller = self._gamepad_axis_command
        axis_map = []
        if isinstance(range_axis, int):
            axis_map.append(range_axis)
        elif isinstance(range_axis, dict):
            for key in self._range.keys():
                axis_map.append(range_axis[key])
        self._gamepad_cfg = {"gamepad": gamepad, "axis_map": axis_map}
        self._gamepad_axis_command_buffer = torch.zeros_like(self._command, device=gs.device)
        self.env.invalidate_trace()   # (as use_external_controller: a recorded step holds the internal generator's launch)

    def _gamepad_axis_command(self, step_count: int) -> torch.Tensor:
        if self._gamepad_cfg is None:
            return self._gamepad_axis_command_buffer
        gamepad, axis_map = self._gamepad_cfg["gamepad"], self._gamepad_cfg["axis_map"]
        cmd = self._gamepad_axis_command_buffer
        ranges = self._ranges()
        for i, axis in enumerate(axis_map):
            if i < len(ranges):
                lo, hi = ranges[i]
                cmd[:, i] = (gamepad.state.axis(axis) + 1) * (hi - lo) / 2 + lo
        return cmd


class VelocityCommandManager(CommandManager):
    """lin_vel_x / lin_vel_y / ang_vel_z command (velocity_command.py:100-125).

    ``standing_probability`` is accepted and — exactly as in the reference, where ``_resample_command`` is
    never reached (quirk q2; command_manager.py:162,170 vs velocity_command.py:194) — has no effect.
    ``debug_visualizer`` is accepted and ignored (viewer-only)."""

    def __init__(self, env, range: dict, resample_time_sec: float = 5.0, standing_probability: float = 0.0,
                 debug_visualizer: bool = False, debug_visualizer_cfg: dict | None = None):
        super().__init__(env, range=range, resample_time_sec=resample_time_sec)
        self.standing_probability = standing_probability
        self.debug_visualizer = debug_visualizer
        self.visualizer_cfg = dict(debug_visualizer_cfg or {})
        self._is_standing_env = torch.zeros(env.num_envs, dtype=torch.bool, device=gs.device)

    def use_gamepad(self, gamepad, lin_vel_y_axis: int = 0, lin_vel_x_axis: int = 1, ang_vel_z_axis: int = 2):
        super().use_gamepad(gamepad, range_axis={"lin_vel_x": lin_vel_x_axis, "lin_vel_y": lin_vel_y_axis, "ang_vel_z": ang_vel_z_axis})


# -- annotation types of the reference (velocity_command.py:17-55), for user code that imports them -------------------------------
from typing import Tuple, TypedDict  # noqa: E402


class VelocityCommandRange(TypedDict):
    lin_vel_x: Tuple[float, float]
    lin_vel_y: Tuple[float, float]
    ang_vel_z: Tuple[float, float]


class VelocityDebugVisualizerConfig(TypedDict, total=False):
    """Options of the debug arrows (drawing is Genesis' viewer: out of scope here; the dict is accepted and kept)."""
    envs_idx: list
    arrow_offset: float
    arrow_radius: float
    arrow_max_length: float
    commanded_color: Tuple[float, float, float, float]
    actual_color: Tuple[float, float, float, float]
