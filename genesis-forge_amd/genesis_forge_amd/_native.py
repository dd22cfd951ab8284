"""
ctypes binding of the C ABI declared in ``include/gf_step.h`` (the drop-in boundary).

The structures below mirror the header field for field; ``check_abi`` compares every
``ctypes.sizeof`` with the library's own ``gf_sizeof`` so a drifted binding fails at load time,
not with a corrupted launch.  There is no CPU fallback: if ``libgf_step.so`` (built by
``__graft_entry__.build()`` / ``make -C genesis-forge_amd/csrc``) is missing, or no ROCm device
is visible, every phase call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

GF_ABI_VERSION = 7
GF_MAX_TERMS = 24
GF_MAX_TERM_TERMS = 16
GF_MAX_OBS_ITEMS = 24
GF_MAX_CONTACT_VIEWS = 4
GF_MAX_COMMAND_VIEWS = 4
GF_MAX_EXT = 16
GF_MAX_LINK_IDS = 32
GF_MAX_RANGES = 8
GF_MAX_OBS_WIDTH = 256
GF_MAX_GAITS = 4
GF_STATS_SHARDS = 64

# opcodes ------------------------------------------------------------------------------------
GF_ACTION_POSITION, GF_ACTION_WITHIN_LIMITS = 0, 1

GF_T_TIMEOUT = 1
GF_T_BAD_ORIENTATION = 2
GF_T_BASE_HEIGHT_BELOW = 3
GF_T_OUT_OF_BOUNDS = 4
GF_T_HAS_CONTACT = 5
GF_T_CONTACT_FORCE = 6
GF_T_CONTACT_FORCE_GRACE = 7
GF_T_EXTERNAL = 8
GF_TERM_FLAG_TIME_OUT = 1

GF_R_IS_ALIVE = 1
GF_R_TERMINATED = 2
GF_R_BASE_HEIGHT = 3
GF_R_DOF_SIMILAR_TO_DEFAULT = 4
GF_R_LIN_VEL_Z_L2 = 5
GF_R_ANG_VEL_XY_L2 = 6
GF_R_FLAT_ORIENTATION_L2 = 7
GF_R_BODY_ACCEL_EXP = 8
GF_R_ACTION_RATE_L2 = 9
GF_R_CMD_TRACK_LIN_VEL = 10
GF_R_CMD_TRACK_ANG_VEL = 11
GF_R_STAND_STILL = 12
GF_R_HAS_CONTACT = 13
GF_R_CONTACT_FORCE = 14
GF_R_FEET_AIR_TIME = 15
GF_R_FEET_SLIDE = 16
GF_R_EXTERNAL = 17
GF_R_GAIT_PHASE = 18
GF_R_FOOT_HEIGHT = 19
GF_RW_FLAG_CMD, GF_RW_FLAG_TERRAIN, GF_RW_FLAG_MAX, GF_RW_FLAG_FIRST_CALL = 1, 2, 4, 8
GF_REWARD_MODE_STEP, GF_REWARD_MODE_EVAL = 0, 1

GF_CMD_STEP, GF_CMD_MASKED, GF_CMD_ALL = 0, 1, 2
(GF_GAIT_OFFSET, GF_GAIT_HEIGHT, GF_GAIT_PERIOD, GF_GAIT_CLOCK, GF_GAIT_TIME, GF_GAIT_PHASE, GF_GAIT_ROW,
 GF_GAIT_OBS_WIDTH) = 0, 4, 5, 6, 14, 15, 16, 14

GF_O_COMMAND = 1
GF_O_ANG_VEL_BODY = 2
GF_O_LIN_VEL_BODY = 3
GF_O_PROJ_GRAVITY = 4
GF_O_DOF_POS = 5
GF_O_DOF_VEL = 6
GF_O_DOF_FORCE = 7
GF_O_ACTIONS = 8
GF_O_RAW_ACTIONS = 9
GF_O_CONTACT_FORCE_NORM = 10
GF_O_EXTERNAL = 11
GF_O_BASE_POS = 12
GF_O_BASE_QUAT = 13

GF_ROT_PROJ_GRAVITY, GF_ROT_LIN_VEL, GF_ROT_ANG_VEL = 0, 1, 2

(GF_PHASE_ACTION, GF_PHASE_CONTACT, GF_PHASE_TERMINATION, GF_PHASE_REWARD, GF_PHASE_COMMAND,
 GF_PHASE_RESET, GF_PHASE_OBSERVE, GF_PHASE_ROTATE, GF_PHASE_SCENE, GF_PHASE_POST, GF_PHASE_TERRAIN, GF_PHASE_GAIT, GF_PHASE_ROLLOUT, GF_PHASE_UNROLL,
 GF_PHASE_ROLLOUT_POLICY, GF_PHASE_GAE, GF_PHASE_COMPACT, GF_PHASE_COUNT) = range(18)

GF_OPT_PROFILE_STRIDE = 1  # gf_set_option: stamp every k-th launch of the profiled phase
GF_OPT_CHAIN = 3           # gf_set_option: 1 (default) = fold runs of per-env phases of a recorded step into phase-chain launches
GF_OPT_GRAPH = 2           # gf_set_option: 1 = recorded steps replay as one hipGraphLaunch; 0 (default: measured faster) = plain launches
GF_OPT_FOLD_CONTACT = 4    # gf_set_option: 1 (default) = the contact ops in front of a fused post-physics op run as that launch's first phase
GF_OPT_POST_VARIANT = 0  # gf_set_option: 0 = interpreter, one wave per tile; 1 = interpreter, four waves; 2 = + static programs (default)

GF_ERRORS = {-1: "GF_E_NULL", -2: "GF_E_RANGE", -3: "GF_E_OPCODE", -4: "GF_E_SLOT", -5: "GF_E_UNSUPPORTED"}

P = C.c_void_p  # every device pointer crosses the ABI as a plain address


class GfEntityView(C.Structure):
    _fields_ = [("pos", P), ("quat", P), ("lin_vel", P), ("ang_vel", P)]


class GfContactView(C.Structure):
    _fields_ = [("contacts", P), ("last_air_time", P), ("current_contact_time", P), ("link_vel", P), ("link_pos", P),
                ("num_links", C.c_int32), ("_pad", C.c_int32)]


class GfCommandView(C.Structure):
    _fields_ = [("command", P), ("width", C.c_int32), ("stride", C.c_int32)]


class GfTerrainView(C.Structure):
    _fields_ = [("height_field", P), ("rows", C.c_int32), ("cols", C.c_int32), ("x_min", C.c_float), ("x_span", C.c_float),
                ("y_min", C.c_float), ("y_span", C.c_float), ("origin_z", C.c_float), ("_pad", C.c_int32)]


class GfTerrainHeightArgs(C.Structure):
    _fields_ = [("num", C.c_int64), ("x", P), ("y", P), ("x_stride", C.c_int64), ("y_stride", C.c_int64),
                ("terrain", GfTerrainView), ("out", P)]


class GfTerm(C.Structure):
    _fields_ = [("op", C.c_int32), ("flags", C.c_int32), ("w", C.c_float), ("p", C.c_float * 4),
                ("i", C.c_int32 * 4), ("row", C.c_int32)]


class GfStepStats(C.Structure):
    _fields_ = [("term_fired", C.c_int32 * GF_MAX_TERM_TERMS), ("reset_count", C.c_int32),
                ("action_flags", C.c_int32), ("contact_flags", C.c_int32), ("resample_count", C.c_int32),
                ("gait_count", C.c_int32 * GF_MAX_GAITS), ("reward_episode_sum", C.c_double * GF_MAX_TERMS)]


class GfActionArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_dofs", C.c_int32), ("mode", C.c_int32), ("check_finite", C.c_int32),
                ("actions_in", P), ("scale", P), ("offset", P), ("clip_lo", P), ("clip_hi", P),
                ("env_actions", P), ("env_last_actions", P), ("episode_length", P), ("targets", P), ("stats", P), ("stats_zero", P),
                ("stats_fold_src", P), ("stats_fold_dst", P), ("stats_last_reset", P)]


class GfContactArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_contacts", C.c_int32), ("num_scene_links", C.c_int32),
                ("num_targets", C.c_int32), ("num_with", C.c_int32), ("has_with_filter", C.c_int32),
                ("track_air_time", C.c_int32), ("_pad", C.c_int32),
                ("force", P), ("position", P), ("link_a", P), ("link_b", P), ("links_quat", P), ("links_vel", P), ("link_vel_out", P), ("links_pos", P), ("link_pos_out", P),
                ("target_link_ids", C.c_int32 * GF_MAX_LINK_IDS), ("with_link_ids", C.c_int32 * GF_MAX_LINK_IDS),
                ("air_time_threshold", C.c_float), ("dt", C.c_float),
                ("contacts", P), ("contact_positions", P), ("position_counts", P),
                ("last_air_time", P), ("current_air_time", P), ("last_contact_time", P), ("current_contact_time", P),
                ("stats", P)]


class GfTerminationArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_terms", C.c_int32), ("entity", GfEntityView),
                ("episode_length", P), ("max_episode_length", P),
                ("contact", GfContactView * GF_MAX_CONTACT_VIEWS), ("ext", P * GF_MAX_EXT),
                ("terminated", P), ("truncated", P), ("term_out", P), ("stats", P),
                ("terms", GfTerm * GF_MAX_TERM_TERMS)]


class GfRewardArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_dofs", C.c_int32), ("num_terms", C.c_int32), ("mode", C.c_int32),
                ("dt", C.c_float), ("logging_enabled", C.c_int32), ("entity", GfEntityView),
                ("dof_pos", P), ("default_dof_pos", P), ("actions", P), ("last_actions", P), ("terminated", P),
                ("contact", GfContactView * GF_MAX_CONTACT_VIEWS), ("command", GfCommandView * GF_MAX_COMMAND_VIEWS),
                ("ext", P * GF_MAX_EXT), ("state", P * 4),
                ("reward", P), ("episode_sums", P), ("episode_seconds", P), ("term_out", P), ("terrain", GfTerrainView),
                ("gait_wave_flags", P), ("terms", GfTerm * GF_MAX_TERMS)]


class GfCommandArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_ranges", C.c_int32), ("mode", C.c_int32), ("resample_steps", C.c_int32),
                ("episode_length", P), ("mask", P), ("mask2", P), ("draws", P),
                ("seed", C.c_uint64), ("stream", C.c_uint64), ("env_offset", C.c_uint32), ("_pad2", C.c_uint32),
                ("lo", C.c_float * GF_MAX_RANGES), ("hi", C.c_float * GF_MAX_RANGES),
                ("command", P), ("stats", P)]


class GfResetArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_dofs", C.c_int32), ("num_reward_terms", C.c_int32), ("num_contact", C.c_int32),
                ("mask", P), ("mask2", P),
                ("env_actions", P), ("env_last_actions", P), ("episode_length", P), ("max_episode_length", P),
                ("base_max_episode_length", C.c_int32), ("max_random_scaling", C.c_float), ("len_draws", P),
                ("episode_sums", P), ("episode_seconds", P), ("reward_log_mask", C.c_uint32), ("reward_logging", C.c_int32),
                ("air_state", (P * 4) * GF_MAX_CONTACT_VIEWS), ("air_links", C.c_int32 * GF_MAX_CONTACT_VIEWS),
                ("scene_dof_pos", P), ("scene_dof_vel", P), ("default_dof_pos", P), ("dof_noise_scale", C.c_float),
                ("dof_draws", P), ("scene_pos", P), ("scene_quat", P), ("quat_stash", P), ("scene_lin_vel", P), ("scene_ang_vel", P),
                ("reset_pos", C.c_float * 3), ("reset_quat", C.c_float * 4), ("set_quat", C.c_int32), ("zero_velocity", C.c_int32),
                ("seed", C.c_uint64), ("stream", C.c_uint64), ("env_offset", C.c_uint32),
                ("spawn_mode", C.c_int32), ("spawn_set_quat", C.c_int32), ("spawn_rot_mask", C.c_int32),
                ("spawn_x_min", C.c_float), ("spawn_x_span", C.c_float), ("spawn_y_min", C.c_float), ("spawn_y_span", C.c_float),
                ("spawn_height_offset", C.c_float), ("spawn_rot_lo", C.c_float * 3), ("spawn_rot_hi", C.c_float * 3),
                ("spawn_draws", P), ("terrain", GfTerrainView), ("stats", P)]


class GfObsItem(C.Structure):
    _fields_ = [("op", C.c_int32), ("width", C.c_int32), ("i0", C.c_int32), ("i1", C.c_int32),
                ("scale", C.c_float), ("noise", C.c_float)]


class GfObservationArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_dofs", C.c_int32), ("num_items", C.c_int32), ("obs_width", C.c_int32),
                ("history_len", C.c_int32), ("history_ring", C.c_int32), ("entity", GfEntityView),
                ("dof_pos", P), ("dof_vel", P), ("dof_force", P), ("targets", P), ("env_actions", P),
                ("contact", GfContactView * GF_MAX_CONTACT_VIEWS), ("command", GfCommandView * GF_MAX_COMMAND_VIEWS),
                ("ext", P * GF_MAX_EXT), ("noise_draws", P), ("seed", C.c_uint64), ("stream", C.c_uint64),
                ("env_offset", C.c_uint32), ("ring_slots", C.c_uint32), ("stale_quat", P), ("stale_mask", P), ("stale_mask2", P), ("prev_obs", P), ("obs", P), ("items", GfObsItem * GF_MAX_OBS_ITEMS)]


class GfRotateArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("what", C.c_int32), ("entity", GfEntityView), ("out", P)]


class GfSynthSceneArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_dofs", C.c_int32), ("num_contacts", C.c_int32), ("num_scene_links", C.c_int32),
                ("dt", C.c_float), ("joint_rate", C.c_float), ("ang_noise", C.c_float), ("lin_noise", C.c_float),
                ("height_target", C.c_float), ("contact_prob", C.c_float), ("contact_force", C.c_float), ("foot_contact_prob", C.c_float),
                ("targets", P), ("pos", P), ("quat", P), ("lin_vel", P), ("ang_vel", P), ("dof_pos", P), ("dof_vel", P),
                ("contact_force_out", P), ("contact_pos_out", P), ("link_a_out", P), ("link_b_out", P),
                ("links_quat_out", P), ("links_vel_out", P), ("links_pos_out", P), ("seed", C.c_uint64), ("tick", C.c_uint64),
                ("env_offset", C.c_uint32), ("foot_link_mask", C.c_uint32)]


class GfOp(C.Structure):
    _fields_ = [("phase", C.c_int32), ("_pad", C.c_int32), ("args", P)]


class GfGaitArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("mode", C.c_int32), ("resample_steps", C.c_int32), ("num_gaits", C.c_int32),
                ("episode_length", P), ("mask", P), ("mask2", P), ("draws", P), ("seed", C.c_uint64), ("stream", C.c_uint64),
                ("env_offset", C.c_uint32), ("fixed_clearance_mask", C.c_int32), ("cum_weight", C.c_float * GF_MAX_GAITS),
                ("gait_offsets", (C.c_float * 4) * GF_MAX_GAITS), ("clearance_lo", C.c_float), ("clearance_hi", C.c_float),
                ("period_lo", C.c_float), ("period_hi", C.c_float), ("dt", C.c_float), ("two_pi", C.c_float),
                ("state", P), ("selected", P), ("wave_flags", P), ("stats", P)]


(GF_PATCH_ACTIONS, GF_PATCH_STREAM, GF_PATCH_COUNTER, GF_PATCH_ROTATE, GF_PATCH_PARAM, GF_PATCH_COPY, GF_PATCH_RING_SLOT,
 GF_PATCH_PARAM_OFFSET) = range(1, 9)


class GfRotor(C.Structure):
    _fields_ = [("cur", C.c_int32), ("count", C.c_int32), ("slot", P * 8)]


class GfRingClock(C.Structure):
    _fields_ = [("calls", C.c_int32), ("length", C.c_int32)]


class GfReplayPatch(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32), ("target", P), ("target2", P), ("aux", P)]


class GfReplay(C.Structure):
    _fields_ = [("ops", P), ("num_ops", C.c_int32), ("num_patches", C.c_int32), ("patches", P), ("rng_stream", P)]


def field_addr(struct, name: str) -> int:
    """Host address of ``struct.name`` (a patch target, see GfReplayPatch)."""
    return C.addressof(struct) + getattr(type(struct), name).offset


class GfStatsCopyArgs(C.Structure):
    _fields_ = [("src", P), ("dst", P), ("event", P)]


GF_OP_STATS_CLEAR, GF_OP_STATS_COPY, GF_OP_POST_PHYSICS, GF_OP_STATS_PACK = 100, 101, 102, 103


class GfStatsPackArgs(C.Structure):
    _fields_ = [("src", P), ("dst", P)]

GF_POST_MAX_CMD, GF_POST_MAX_OBS, GF_POST_MAX_GAIT = 2, 2, 1
GF_POST_TERMINATION_DONE = 1   # GfPostRefs.flags
GF_POST_OBSERVE_ONLY = 2
GF_POST_NO_RESET = 4


class GfRolloutArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("obs_width", C.c_int32), ("obs", P), ("reward", P), ("terminated", P), ("truncated", P),
                ("obs_out", P), ("reward_out", P), ("done_out", P)]


class GfHistoryUnrollArgs(C.Structure):
    _fields_ = [("ring", P), ("out", P), ("out2", P), ("num_envs", C.c_int64), ("frame_width", C.c_int32), ("history_len", C.c_int32),
                ("ring_slot", C.c_int32), ("_pad", C.c_int32)]


class GfRolloutPolicyArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_actions", C.c_int32), ("actions", P), ("values", P), ("log_prob", P), ("mu", P), ("sigma", P),
                ("actions_out", P), ("values_out", P), ("log_prob_out", P), ("mu_out", P), ("sigma_out", P),
                ("time_outs", P), ("reward_row", P), ("gamma", C.c_float), ("_pad", C.c_int32)]


class GfGaeArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("num_steps", C.c_int32), ("rewards", P), ("values", P), ("dones", P), ("last_values", P),
                ("gamma", C.c_float), ("lam", C.c_float), ("returns", P), ("advantages", P), ("moments", P),
                ("normalize", C.c_int32), ("_pad", C.c_int32)]


class GfCompactArgs(C.Structure):
    _fields_ = [("num_envs", C.c_int64), ("mask", P), ("mask2", P), ("ids_out", P), ("count_out", P), ("block_counts", P),
                ("wait", C.c_int32), ("_pad", C.c_int32)]


class GfPostRefs(C.Structure):
    _fields_ = [("termination", P), ("reward", P), ("reset", P), ("num_command", C.c_int32), ("num_observe", C.c_int32),
                ("command_step", P * GF_POST_MAX_CMD), ("command_reset", P * GF_POST_MAX_CMD), ("observe", P * GF_POST_MAX_OBS),
                ("num_gait", C.c_int32), ("flags", C.c_int32), ("gait_step", P * GF_POST_MAX_GAIT), ("gait_reset", P * GF_POST_MAX_GAIT),
                ("gait_flags_next", P * GF_POST_MAX_GAIT), ("rollout", P)]

ABI_STRUCTS = [GfStepStats, GfActionArgs, GfContactArgs, GfTerminationArgs, GfRewardArgs, GfCommandArgs,
               GfResetArgs, GfObservationArgs, GfRotateArgs, GfSynthSceneArgs, GfTerm, GfObsItem, GfTerrainView, GfTerrainHeightArgs, GfGaitArgs, GfContactView, GfCommandView,
               GfPostRefs, GfRolloutArgs, GfHistoryUnrollArgs, GfRolloutPolicyArgs, GfGaeArgs, GfCompactArgs]

PHASE_FUNCS = {
    "action_step": GfActionArgs,
    "contact_step": GfContactArgs,
    "termination_step": GfTerminationArgs,
    "reward_step": GfRewardArgs,
    "command_step": GfCommandArgs,
    "masked_reset": GfResetArgs,
    "observe": GfObservationArgs,
    "entity_rotate": GfRotateArgs,
    "synth_scene_step": GfSynthSceneArgs,
    "terrain_height": GfTerrainHeightArgs,
    "gait_step": GfGaitArgs,
    "rollout_write": GfRolloutArgs,
    "history_unroll": GfHistoryUnrollArgs,
    "rollout_policy_write": GfRolloutPolicyArgs,
    "gae": GfGaeArgs,
    "done_compact": GfCompactArgs,
}


PHASE_OF_FN = {
    "action_step": GF_PHASE_ACTION, "contact_step": GF_PHASE_CONTACT, "termination_step": GF_PHASE_TERMINATION,
    "reward_step": GF_PHASE_REWARD, "command_step": GF_PHASE_COMMAND, "masked_reset": GF_PHASE_RESET,
    "observe": GF_PHASE_OBSERVE, "entity_rotate": GF_PHASE_ROTATE, "synth_scene_step": GF_PHASE_SCENE,
    "terrain_height": GF_PHASE_TERRAIN, "gait_step": GF_PHASE_GAIT, "rollout_write": GF_PHASE_ROLLOUT,
    "history_unroll": GF_PHASE_UNROLL, "rollout_policy_write": GF_PHASE_ROLLOUT_POLICY, "gae": GF_PHASE_GAE, "done_compact": GF_PHASE_COMPACT,
}


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libgf_step.so")


class GfError(RuntimeError):
    """Raised when a native phase call returns a non-zero status (include/gf_step.h error convention)."""


def check_abi(lib, prefix: str = "gf_") -> None:
    sizeof = getattr(lib, prefix + "sizeof")
    sizeof.restype = C.c_int
    sizeof.argtypes = [C.c_int]
    ver = getattr(lib, prefix + "abi_version")
    ver.restype = C.c_int
    if ver() != GF_ABI_VERSION:
        raise GfError(f"ABI version mismatch: library {ver()} vs binding {GF_ABI_VERSION}")
    for idx, st in enumerate(ABI_STRUCTS):
        n = sizeof(idx)
        if n != C.sizeof(st):
            raise GfError(f"ABI drift: sizeof({st.__name__}) is {n} in the library, {C.sizeof(st)} in the binding")


def data_ptr(t) -> Optional[int]:
    """Address of a tensor's storage (None stays NULL)."""
    if t is None:
        return None
    return t.data_ptr()


class Backend:
    """What the managers call.  One implementation ships: :class:`HipBackend`."""

    name = "abstract"
    device_type = "cuda"

    tracer = None  # set by _trace.StepTrace while it records a step
    graph_enabled = False   # GF_GRAPH=1 (HipBackend): recorded steps replay as one hipGraphLaunch (measured slower, see gf_step.h)

    def _note_call(self, args) -> None:
        """A phase call outside a recorded step's replay.  If its descriptor belongs to a live recorded step (`watched`), that
        step's frozen copy of it is no longer what the ordinary path would launch — a manager method called between steps
        (``resample_command([…])``, ``reset([…])``) has refilled it — and the recording must go (`dirty`, checked by
        StepTrace.fresh before every replay)."""
        w = self.__dict__.get("watched")
        if w:
            a = C.addressof(args)
            if a in w:
                self.__dict__.setdefault("dirty", set()).add(a)

    def call(self, fn: str, args, owner=None) -> None:  # pragma: no cover - interface
        raise NotImplementedError

    def stats_clear(self, stats_ptr: int) -> None:  # pragma: no cover - interface
        raise NotImplementedError

    def run_ops(self, ops, n: int) -> None:  # pragma: no cover - interface
        raise NotImplementedError

    def replay_step(self, replay, actions_ptr: int, params, num_params: int) -> None:  # pragma: no cover - interface
        """Apply a recorded step's patch table, then replay its ops (gf_replay_step)."""
        raise NotImplementedError

    def event_create(self):
        return None

    def event_synchronize(self, ev) -> None:
        pass

    def post_check(self, refs) -> bool:
        """True when gf_post_physics_step can fuse the phases ``refs`` points at."""
        return False

    def stats_pack(self, src_ptr: int, dst_ptr: int) -> None:  # pragma: no cover - interface
        raise NotImplementedError

    def stats_last_reset(self, rows_ptr: int, num_rows: int, dst_ptr: int) -> None:  # pragma: no cover - interface
        raise NotImplementedError

    def check_tensor(self, t, what: str = "tensor") -> None:
        if t is None:
            return
        if t.device.type != self.device_type:
            raise GfError(f"{what} lives on {t.device}, but the {self.name} backend needs {self.device_type} tensors")
        if not t.is_contiguous():
            raise GfError(f"{what} must be contiguous")


class HipBackend(Backend):
    """ctypes front of libgf_step.so; launches on torch's current HIP stream."""

    name = "hip"
    device_type = "cuda"

    def __init__(self, path: Optional[str] = None):
        path = path or lib_path()
        if not os.path.exists(path):
            raise GfError(
                f"{path} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C genesis-forge_amd/csrc).  genesis_forge_amd has no CPU fallback.")
        import torch  # plumbing only: makes sure torch's HIP runtime is the one in the process

        self._torch = torch
        self._gpu_checked = False
        self._raw_stream = None
        self.lib = C.CDLL(path)
        check_abi(self.lib, "gf_")
        self._fn = {}
        for name, st in PHASE_FUNCS.items():
            f = getattr(self.lib, "gf_" + name)
            f.restype = C.c_int
            f.argtypes = [C.POINTER(st), C.c_void_p]
            self._fn[name] = f
        self.lib.gf_stats_clear.restype = C.c_int
        self.lib.gf_stats_clear.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.gf_error_string.restype = C.c_char_p
        self.lib.gf_error_string.argtypes = [C.c_int]
        self.lib.gf_build_info.restype = C.c_char_p
        self.lib.gf_run_ops.restype = C.c_int
        self.lib.gf_run_ops.argtypes = [C.POINTER(GfOp), C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        self.lib.gf_replay_step.restype = C.c_int
        self.lib.gf_replay_step.argtypes = [C.POINTER(GfReplay), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        self._failed = C.c_int(-1)
        self.lib.gf_run_ops_graph.restype = C.c_int
        self.lib.gf_run_ops_graph.argtypes = [C.POINTER(C.c_void_p), C.POINTER(GfOp), C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        self.lib.gf_graph_destroy.restype = C.c_int
        self.lib.gf_graph_destroy.argtypes = [C.POINTER(C.c_void_p)]
        if os.environ.get("GF_NO_CHAIN", "0") == "1":
            self.lib.gf_set_option(GF_OPT_CHAIN, 0)
        if os.environ.get("GF_GRAPH", "0") == "1":   # opt-in: measured slower than plain launches (gf_step.h, GF_OPT_GRAPH)
            self.lib.gf_set_option(GF_OPT_GRAPH, 1)
            self.graph_enabled = True
        self.lib.gf_stats_pack.restype = C.c_int
        self.lib.gf_stats_pack.argtypes = [C.POINTER(GfStatsPackArgs), C.c_void_p]
        self.lib.gf_stats_last_reset.restype = C.c_int
        self.lib.gf_stats_last_reset.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.lib.gf_post_physics_check.restype = C.c_int
        self.lib.gf_post_physics_check.argtypes = [C.POINTER(GfPostRefs)]
        self.lib.gf_post_physics_step_contacts.restype = C.c_int
        self.lib.gf_post_physics_step_contacts.argtypes = [C.POINTER(GfPostRefs), C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
        self.lib.gf_post_physics_describe.restype = C.c_int
        self.lib.gf_post_physics_describe.argtypes = [C.POINTER(GfPostRefs), C.c_char_p, C.c_int]
        self.lib.gf_post_program_register.restype = C.c_int
        self.lib.gf_post_program_register.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        self.lib.gf_event_create.restype = C.c_void_p
        self.lib.gf_event_synchronize.restype = C.c_int
        self.lib.gf_event_synchronize.argtypes = [C.c_void_p]
        self.lib.gf_set_option.restype = C.c_int
        self.lib.gf_set_option.argtypes = [C.c_int, C.c_int]
        self.lib.gf_profile_begin.restype = C.c_int
        self.lib.gf_profile_begin.argtypes = [C.c_int, C.c_int]
        variant = os.environ.get("GF_POST_VARIANT")
        if variant is not None:
            self.set_option(GF_OPT_POST_VARIANT, int(variant))
        self.lib.gf_profile_end.restype = C.c_int
        self.lib.gf_profile_end.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int)]

    def _stream(self) -> int:
        torch = self._torch
        if not self._gpu_checked:  # once: every later call is on the hot path of a step
            if not torch.cuda.is_available():
                raise GfError("no ROCm device visible: genesis_forge_amd runs its manager phases as HIP kernels only")
            self._gpu_checked = True
            # torch.cuda.current_stream() builds a Stream object (≈ 3.4 µs per call — a sixth of a 20 µs step); the raw
            # handle of the current stream of the current device is what the C ABI wants
            self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
            self._get_device = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)
        if self._raw_stream is not None:
            return self._raw_stream(self._get_device())   # (torch.cuda.current_device() goes through _lazy_init: ~0.5 µs more)
        return torch.cuda.current_stream().cuda_stream

    def _raise(self, fn: str, rc: int):
        msg = self.lib.gf_error_string(rc).decode()
        raise GfError(f"gf_{fn} failed: {GF_ERRORS.get(rc, rc)} ({msg})")

    def call(self, fn: str, args, owner=None) -> None:
        if self.tracer is not None:
            self.tracer.record(fn, args, owner)
        self._note_call(args)
        rc = self._fn[fn](C.byref(args), self._stream())
        if rc != 0:
            self._raise(fn, rc)

    def run_ops(self, ops, n: int) -> None:
        failed = C.c_int(-1)
        rc = self.lib.gf_run_ops(ops, n, self._stream(), C.byref(failed))
        if rc != 0:
            self._raise(f"run_ops[op {failed.value}]", rc)

    def replay_step(self, replay, actions_ptr: int, params, num_params: int) -> None:
        rc = self.lib.gf_replay_step(replay, actions_ptr, params, num_params, self._stream(), self._failed)
        if rc != 0:
            self._raise(f"replay_step[op {self._failed.value}]", rc)

    def run_ops_graph(self, cache, ops, n: int) -> None:
        """``cache``: a ctypes c_void_p owned by the caller (the recorded step); see gf_run_ops_graph."""
        failed = C.c_int(-1)
        rc = self.lib.gf_run_ops_graph(C.byref(cache), ops, n, self._stream(), C.byref(failed))
        if rc != 0:
            self._raise(f"run_ops_graph[op {failed.value}]", rc)

    def graph_destroy(self, cache) -> None:
        self.lib.gf_graph_destroy(C.byref(cache))

    def post_check(self, refs) -> bool:
        return self.lib.gf_post_physics_check(C.byref(refs)) == 0

    def post_step_contacts(self, refs, contact_args: list) -> int:
        """gf_post_physics_step_contacts: the fused post-physics launch with these ContactManagers' step as its first phase.  Returns the
        status (0, or GF_E_UNSUPPORTED = -5 when the managers cannot be folded) — raw, for callers that fall back themselves."""
        arr = (C.c_void_p * max(1, len(contact_args)))(*[C.addressof(a) for a in contact_args])
        return self.lib.gf_post_physics_step_contacts(C.byref(refs), arr, len(contact_args), self._stream())

    def post_describe(self, refs) -> str:
        buf = C.create_string_buffer(4096)
        rc = self.lib.gf_post_physics_describe(C.byref(refs), buf, len(buf))
        if rc != 0:
            self._raise("post_physics_describe", rc)
        return buf.value.decode()

    def register_program(self, plugin_path: str) -> int:
        """Register a static program compiled at run time (genesis_forge_amd/_programs.py); returns its program id."""
        pid = C.c_int(-1)
        rc = self.lib.gf_post_program_register(plugin_path.encode(), C.byref(pid))
        if rc != 0:
            self._raise(f"post_program_register({plugin_path})", rc)
        return pid.value

    def stats_pack(self, src_ptr: int, dst_ptr: int) -> None:
        a = GfStatsPackArgs()
        a.src, a.dst = src_ptr, dst_ptr
        rc = self.lib.gf_stats_pack(C.byref(a), self._stream())
        if rc != 0:
            self._raise("stats_pack", rc)

    def stats_last_reset(self, rows_ptr: int, num_rows: int, dst_ptr: int) -> None:
        rc = self.lib.gf_stats_last_reset(rows_ptr, num_rows, dst_ptr, self._stream())
        if rc != 0:
            self._raise("stats_last_reset", rc)

    def event_create(self):
        ev = self.lib.gf_event_create()
        if not ev:
            raise GfError("gf_event_create failed")
        return ev

    def event_synchronize(self, ev) -> None:
        rc = self.lib.gf_event_synchronize(ev)
        if rc != 0:
            self._raise("event_synchronize", rc)

    def stats_clear(self, stats_ptr: int) -> None:
        rc = self.lib.gf_stats_clear(stats_ptr, self._stream())
        if rc != 0:
            self._raise("stats_clear", rc)

    def set_option(self, option: int, value: int) -> None:
        if option == GF_OPT_GRAPH:
            self.graph_enabled = bool(value)
        rc = self.lib.gf_set_option(option, value)
        if rc != 0:
            self._raise("set_option", rc)

    def profile_begin(self, phase: int, max_samples: int) -> None:
        rc = self.lib.gf_profile_begin(phase, max_samples)
        if rc != 0:
            self._raise("profile_begin", rc)

    def profile_end(self):
        tot, cnt = C.c_double(0.0), C.c_int(0)
        rc = self.lib.gf_profile_end(C.byref(tot), C.byref(cnt))
        if rc != 0:
            self._raise("profile_end", rc)
        return tot.value, cnt.value

    def build_info(self) -> str:
        return self.lib.gf_build_info().decode()


_backend: Optional[Backend] = None


def get_backend() -> Backend:
    """The process-wide backend; created on first use.  Raises if the HIP library is not built."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def set_backend(b: Optional[Backend]) -> None:
    """Replace the backend (tests inject an oracle-backed one; ``None`` resets to lazy HIP)."""
    global _backend
    _backend = b
