"""
GaitCommandManager — the periodic-gait command manager of the reference's gait_trainer example
(examples/gait_trainer/gait_command_manager.py; "Sim-to-Real Learning of All Common Bipedal Gaits via Periodic Reward
Composition", Siekmann et al. 2020), with the same constructor, attributes, curriculum methods and reward methods, on
the native path.

In the reference this is user-level Python on top of ``CommandManager``: every step it runs a ``nonzero()`` host sync, a
``torch.multinomial`` with per-gait masked scatters, four ``(gait_selected == i).sum()`` reductions for the log and ≈ 30
elementwise launches for the phase clock; its two reward methods are ≈ 40 more.  Here:

* ``step`` / ``reset`` / ``resample_command`` are one ``gf_gait_step`` launch over a ``[N,16]`` state buffer (64-byte rows:
  foot_offset ×4, foot_height, gait_period, clock_input ×8, gait_time, gait_phase).  ``foot_offset`` … ``clock_input`` are
  views into it, ``command`` is columns 0..5 and ``observation()`` columns 0..13, so the observation kernel reads it in place;
* ``gait_phase_reward`` / ``foot_height_reward`` carry ``_gf_spec`` and compile to the ``GF_R_GAIT_PHASE`` /
  ``GF_R_FOOT_HEIGHT`` opcodes of the fused reward kernel when a ContactManager tracking the four feet is available
  (forces, link velocities and link positions come from its per-link buffers); otherwise they evaluate in torch exactly
  like the reference's code;
* the per-gait env counts of ``_log_metrics`` are wave-ballot counts in the step statistics block (summed across ranks by
  the logging all-reduce), read lazily.

The reference's unmodified file also runs on this package as an ordinary user manager (tests/test_examples.py); this class is
the fast path.  Gamepad control (``use_gamepad``) is human-input I/O and out of scope.
"""
from __future__ import annotations

import math
from typing import Literal, Optional, TypedDict

import torch

from .. import _native as nat
from .. import gs
from ._program import TermSpec, eval_reward_spec
from .action import _tag
from .command import CommandManager

GAIT_PERIOD_RANGE = [0.3, 0.6]
FOOT_CLEARANCE_RANGE = [0.04, 0.12]

GaitName = Literal["walk", "trot", "pronk", "pace", "bound", "canter"]
FootName = Literal["FL", "FR", "RL", "RR"]

#: phase offset of each foot's ground contact within the gait cycle (gait_command_manager.py:26-64; the reference keeps
#: "walk" and "canter" commented out, so four gaits are active)
GAIT_OFFSETS: dict = {
    "trot": {"FL": 0.0, "FR": 0.5, "RL": 0.5, "RR": 0.0},
    "pace": {"FL": 0.5, "FR": 0.0, "RL": 0.5, "RR": 0.0},
    "bound": {"FL": 0.0, "FR": 0.0, "RL": 0.5, "RR": 0.5},
    "pronk": {"FL": 0.0, "FR": 0.0, "RL": 0.0, "RR": 0.0},
}
FIXED_CLEARANCE_GAITS = ("pronk", "bound")  # :366-368


class FootNames(TypedDict):
    FL: str
    FR: str
    RL: str
    RR: str


class _RewardMethod:
    """A reward method of the manager: callable like the reference's bound method, and compilable (``_gf_spec``)."""

    def __init__(self, mgr, name: str, spec, torch_impl):
        self._mgr, self._spec, self._impl = mgr, spec, torch_impl
        self.__name__ = name
        self.__qualname__ = f"GaitCommandManager.{name}"
        self.__doc__ = torch_impl.__doc__

    def _gf_spec(self, env, **params):
        return self._spec(env, **params)

    def __call__(self, env, **params):
        spec = self._spec(env, **params)
        if spec is None:
            return self._impl(env, **params)
        return eval_reward_spec(env, spec)


class GaitCommandManager(CommandManager):
    """Gait parameters (per-foot phase offsets, foot clearance, gait period) and the periodic clock for a quadruped
    (ctor as gait_command_manager.py:85-125)."""

    _gf_native_gait = True

    def __init__(self, env, foot_names: FootNames, resample_time_sec: float = 5.0, robot_entity_attr: str = "robot"):
        super().__init__(env, range={}, resample_time_sec=resample_time_sec)
        if len(GAIT_OFFSETS) > nat.GF_MAX_GAITS:
            raise ValueError(f"at most {nat.GF_MAX_GAITS} gaits")
        self._robot_entity_attr = robot_entity_attr
        self._foot_names = foot_names
        self.foot_links: list = []
        self._gamepad = None
        # initial ranges — widened by the curriculum (:101-108)
        self._num_gaits = 1
        self._gait_period_range = [(GAIT_PERIOD_RANGE[0] + GAIT_PERIOD_RANGE[1]) / 2] * 2
        self._foot_clearance_range = [FOOT_CLEARANCE_RANGE[0]] * 2
        self._all_gaits_learned = False
        n = env.num_envs
        self._state = torch.zeros((n, nat.GF_GAIT_ROW), device=gs.device, dtype=torch.float32)
        self._gait_selected = torch.zeros(n, dtype=torch.long, device=gs.device)
        # per block of 64 envs: bit 2f / 2f+1 = some env has foot f in swing / stance (GfGaitArgs.wave_flags); the all-zero
        # initial state is "swing" for every foot.  Padded with zeros to whole 32-bit words.
        blocks = (n + 63) // 64
        flags = torch.zeros((blocks + 3) // 4 * 4, dtype=torch.uint8, device=gs.device)
        flags[:blocks] = 0x55
        #: two buffers: the fused post-physics launch reads the current one and writes the bytes of the state it leaves into the
        #: other, then they swap (GfPostRefs.gait_flags_next); the rotor is shared with a recorded step's native patch table
        self._flag_bufs = [flags, flags.clone()]
        self._flags_rotor = nat.GfRotor()
        self._flags_rotor.cur, self._flags_rotor.count = 0, 2
        for i, b in enumerate(self._flag_bufs):
            self._flags_rotor.slot[i] = b.data_ptr()
        #: reproduce the reference's env-0 index-list quirk in gait_phase_reward (see GF_R_GAIT_PHASE in gf_step.h)
        self.reference_env0_quirk = True
        self._gait_args = {m: nat.GfGaitArgs() for m in (nat.GF_CMD_STEP, nat.GF_CMD_MASKED, nat.GF_CMD_ALL)}
        self._feet_cm = None
        self.gait_phase_reward = _RewardMethod(self, "gait_phase_reward", self._spec_gait_phase, self._gait_phase_reward_torch)
        self.foot_height_reward = _RewardMethod(self, "foot_height_reward", self._spec_foot_height, self._foot_height_reward_torch)

    @property
    def _wave_flags(self) -> torch.Tensor:
        """The swing / stance bytes of the current state."""
        return self._flag_bufs[self._flags_rotor.cur]

    @property
    def _wave_flags_next(self) -> torch.Tensor:
        return self._flag_bufs[1 - self._flags_rotor.cur]

    # -- buffers: views into the state rows (attribute names of gait_command_manager.py:110-125) ------------------------
    @property
    def foot_offset(self) -> torch.Tensor:
        return self._state[:, nat.GF_GAIT_OFFSET:nat.GF_GAIT_OFFSET + 4]

    @property
    def foot_height(self) -> torch.Tensor:
        return self._state[:, nat.GF_GAIT_HEIGHT:nat.GF_GAIT_HEIGHT + 1]

    @property
    def gait_period(self) -> torch.Tensor:
        return self._state[:, nat.GF_GAIT_PERIOD:nat.GF_GAIT_PERIOD + 1]

    @property
    def clock_input(self) -> torch.Tensor:
        return self._state[:, nat.GF_GAIT_CLOCK:nat.GF_GAIT_CLOCK + 8]

    @property
    def gait_time(self) -> torch.Tensor:
        return self._state[:, nat.GF_GAIT_TIME:nat.GF_GAIT_TIME + 1]

    @property
    def gait_phase(self) -> torch.Tensor:
        return self._state[:, nat.GF_GAIT_PHASE:nat.GF_GAIT_PHASE + 1]

    @property
    def command(self) -> torch.Tensor:
        """foot_offset(4) | foot_height | gait_period  (:127-141)"""
        return self._state[:, :nat.GF_GAIT_PERIOD + 1]

    def observation(self, env) -> torch.Tensor:
        """command(6) | clock_input(8)  (:257-268) — read in place by the observation kernel."""
        return _tag(self._state[:, :nat.GF_GAIT_OBS_WIDTH], ("cmd", self))

    def _gf_command_view(self, v: nat.GfCommandView, args=None) -> None:
        v.command, v.width, v.stride = self._state.data_ptr(), nat.GF_GAIT_OBS_WIDTH, nat.GF_GAIT_ROW
        if args is not None and hasattr(args, "gait_wave_flags"):
            args.gait_wave_flags = self._wave_flags.data_ptr() if self.reference_env0_quirk else None

    # -- curriculum (:146-180) ------------------------------------------------------------------------------------------
    def increment_num_gaits(self):
        if self._all_gaits_learned:
            return
        if self._num_gaits == len(GAIT_OFFSETS):
            self._all_gaits_learned = True
            print("🎯 All gaits learned! Switching to uniform sampling.")
        else:
            self._num_gaits = min(self._num_gaits + 1, len(GAIT_OFFSETS))

    def increment_gait_period_range(self):
        self._gait_period_range[0] = max(self._gait_period_range[0] - 0.05, GAIT_PERIOD_RANGE[0])
        self._gait_period_range[1] = min(self._gait_period_range[1] + 0.05, GAIT_PERIOD_RANGE[1])

    def increment_foot_clearance_range(self):
        self._foot_clearance_range[0] = max(self._foot_clearance_range[0] - 0.01, FOOT_CLEARANCE_RANGE[0])
        self._foot_clearance_range[1] = min(self._foot_clearance_range[1] + 0.01, FOOT_CLEARANCE_RANGE[1])

    # -- lifecycle ------------------------------------------------------------------------------------------------------
    def build(self):
        """Resolve the foot links (:213-220)."""
        super().build()
        robot = getattr(self.env, self._robot_entity_attr)
        self.foot_links = [robot.get_link(self._foot_names[key]) for key in ("FL", "FR", "RL", "RR")]

    def _cfg_key(self) -> tuple:
        """What the curriculum (or the user) can change between launches (:146-180, resample_time_sec)."""
        return (self._num_gaits, self._all_gaits_learned, self._gait_period_range[0], self._gait_period_range[1],
                self._foot_clearance_range[0], self._foot_clearance_range[1], self._resample_steps)

    def _fill(self, a: nat.GfGaitArgs, mode: int) -> None:
        """Everything the curriculum can change is re-read on every launch (:195-211, 347-399); the descriptor is only
        rewritten when one of those values actually changed."""
        env = self.env
        key = self._cfg_key()
        if getattr(a, "_gf_key", None) == key:
            a.wave_flags = self._wave_flags.data_ptr()   # the current one of the two swing / stance buffers
            return
        a._gf_key = key
        a.wave_flags = self._wave_flags.data_ptr()
        a.num_envs, a.mode, a.resample_steps = env.num_envs, mode, self._resample_steps
        g = self._num_gaits
        a.num_gaits = g
        # torch.arange(g).exp() / sum, or uniform once every gait is learned (:387-394), in f32 on the host so that every
        # rank and the oracle see the same thresholds
        w = torch.ones(g, dtype=torch.float32) if self._all_gaits_learned else torch.arange(g, dtype=torch.float32).exp()
        w /= w.sum()
        cum = torch.cumsum(w, 0)
        names = list(GAIT_OFFSETS.keys())
        mask = 0
        for k in range(nat.GF_MAX_GAITS):
            a.cum_weight[k] = float(cum[k]) if k < g else 2.0
            if k < len(names):
                for f, foot in enumerate(("FL", "FR", "RL", "RR")):
                    a.gait_offsets[k][f] = GAIT_OFFSETS[names[k]][foot]
                if names[k] in FIXED_CLEARANCE_GAITS:
                    mask |= 1 << k
        a.fixed_clearance_mask = mask
        a.clearance_lo, a.clearance_hi = self._foot_clearance_range
        a.period_lo, a.period_hi = self._gait_period_range
        a.dt, a.two_pi = float(env.dt), 2 * math.pi
        a.seed, a.env_offset = env._rng_seed, env.env_offset
        a.state, a.selected = self._state.data_ptr(), self._gait_selected.data_ptr()
        a.episode_length = env.episode_length.data_ptr()

    def _launch_gait(self, mode: int, mask=None, mask2=None, draws_key: Optional[str] = None) -> None:
        env = self.env
        a = self._gait_args[mode]
        self._fill(a, mode)
        a.mask = None if mask is None else mask.data_ptr()
        a.mask2 = None if mask2 is None else mask2.data_ptr()
        draws = env.take_draws(draws_key) if draws_key else None
        self._keep = (draws, mask, mask2)
        a.draws = None if draws is None else draws.data_ptr()
        a.stream = env.next_stream()
        a.stats = env.stats.ptr if mode == nat.GF_CMD_STEP else None
        env.backend.call("gait_step", a, owner=self)

    def step(self):
        """Resample due envs, log the gait distribution, advance the phase clock (:222-239) — one launch."""
        if not self.enabled or self._gamepad is not None:
            return
        self._launch_gait(nat.GF_CMD_STEP, draws_key=f"gait:{self._index}")
        self._log_metrics()

    def reset(self, env_ids=None):
        """Resample and zero the clock of the given envs (:241-255)."""
        if not self.enabled:
            return
        self.resample_command(env_ids)

    def resample_command(self, env_ids):
        if self._gamepad is not None:
            return
        if env_ids is None:
            self._launch_gait(nat.GF_CMD_ALL, draws_key=f"gait_reset:{self._index}")
        else:
            self._launch_gait(nat.GF_CMD_MASKED, mask=self.env._ids_to_mask(env_ids), draws_key=f"gait_reset:{self._index}")

    def _can_fuse_reset(self) -> bool:
        return True

    def _fill_reset(self, a) -> None:
        pass  # the gait state is reset by its own masked launch (_after_fused_reset)

    def _after_fused_reset(self, mask, mask2) -> None:
        if self.enabled and self._gamepad is None:
            self._launch_gait(nat.GF_CMD_MASKED, mask=mask, mask2=mask2, draws_key=f"gait_reset:{self._index}")

    def _trace_patch(self, args):
        mode = args.mode

        def patch(_actions, a=args, self=self, mode=mode):
            self._fill(a, mode)
            if mode == nat.GF_CMD_STEP:
                self._log_metrics()

        return patch

    def _trace_native(self, args) -> list:
        """The launch's Philox stream as a native patch (what `_launch_gait` draws with env.next_stream())."""
        return [nat.GfReplayPatch(nat.GF_PATCH_STREAM, 0, nat.field_addr(args, "stream"), None, None)]

    def use_gamepad(self, gamepad):
        raise NotImplementedError("gamepad HID input is outside the manager-step pipeline (SURVEY.md §2 row 19)")

    # -- logging (:430-441) ---------------------------------------------------------------------------------------------
    def _log_metrics(self):
        log = self.env._extras[self.env.extras_logging_key]
        log["Metrics / num_gaits"] = self._num_gaits
        names = list(GAIT_OFFSETS.keys())

        def fill(st, out, names=names):
            for i, name in enumerate(names):
                out[f"Metrics / gait_{name}_envs"] = int(st.gait_count[i])

        if hasattr(log, "add_filler"):
            log.add_filler(fill)

    # -- rewards --------------------------------------------------------------------------------------------------------
    def _feet_contact_manager(self, preferred=None):
        """A ContactManager whose tracked links include the four feet: its per-link force / velocity / position buffers feed
        the gait reward opcodes.  Returns (manager, packed row of each foot) or (None, 0)."""
        cands = ([preferred] if preferred is not None else []) + [c for c in self.env.managers["contact"] if c is not preferred]
        for cm in cands:
            ids = cm.link_ids.tolist() if cm.link_ids is not None else []
            try:
                rows = [ids.index(link.idx) for link in self.foot_links]
            except ValueError:
                continue
            if getattr(cm, "_entity_attr", "robot") != self._robot_entity_attr or cm._has_with_filter:
                continue
            packed = 0
            for f, r in enumerate(rows):
                packed |= (r & 0xFF) << (8 * f)
            return cm, packed
        return None, 0

    def _spec_gait_phase(self, env, contact_manager):
        cm, packed = self._feet_contact_manager(contact_manager)
        if cm is not contact_manager or not self.foot_links:
            return None
        return TermSpec(nat.GF_R_GAIT_PHASE, p=[0.0, 2 * math.pi, math.pi], i=[0, 0, packed], contact={0: cm}, cmd={1: self},
                        link_vel=True)

    def _spec_foot_height(self, env, sensitivity: float = 0.1):
        cm, packed = self._feet_contact_manager()
        if cm is None or not self.foot_links:
            return None
        return TermSpec(nat.GF_R_FOOT_HEIGHT, p=[float(sensitivity)], i=[0, 0, packed], contact={0: cm}, cmd={1: self},
                        link_vel=True, link_pos=True)

    def _foot_height_reward_torch(self, env, sensitivity: float = 0.1) -> torch.Tensor:
        """Reward the feet for reaching the target height during the swing phase (:278-293)."""
        link_idx = [f.idx_local for f in self.foot_links]
        foot_vel = env.robot.get_links_vel(links_idx_local=link_idx)
        foot_pos = env.robot.get_links_pos(links_idx_local=link_idx)
        foot_vel_xy_norm = torch.norm(foot_vel[:, :, :2], dim=-1)
        clearance_error = torch.sum(foot_vel_xy_norm * torch.square(foot_pos[:, :, 2] - self.foot_height), dim=-1)
        return torch.exp(-clearance_error / sensitivity)

    def _gait_phase_reward_torch(self, env, contact_manager) -> torch.Tensor:
        """Reward the feet for being in the correct phase (:295-345)."""
        quad = None
        for foot_idx in range(4):
            link = self.foot_links[foot_idx]
            force = torch.norm(contact_manager.get_contact_forces(link.idx), dim=-1).view(-1, 1)
            velocity = torch.norm(link.get_vel(), dim=-1).view(-1, 1)
            phi = (self.gait_phase + self.foot_offset[:, foot_idx].unsqueeze(1)) % 1.0
            phi = phi * (2 * torch.pi)
            swing = (phi >= 0.0) & (phi < torch.pi)
            stance = (phi >= torch.pi) & (phi < 2 * torch.pi)
            if self.reference_env0_quirk:  # `.nonzero().flatten()` of the [N,1] masks puts index 0 into both lists (:335-343)
                any_swing, any_stance = bool(swing.any()), bool(stance.any())
                if any_stance:
                    swing[0], stance[0] = False, True
                elif any_swing:
                    swing[0], stance[0] = True, False
            foot = (-stance.to(torch.float32)) * velocity + (-swing.to(torch.float32)) * force
            quad = foot.flatten() if quad is None else quad + foot.flatten()
        return torch.exp(quad)
