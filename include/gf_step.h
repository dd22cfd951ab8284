/*
 * gf_step.h — C ABI of the MI355X-native ManagedEnvironment manager-step pipeline.
 *
 * One entry point per phase of the reference's ManagedEnvironment.step()
 * (/root/reference/genesis_forge/managed_env.py:274-334).  Every function takes a POD
 * descriptor of caller-owned device buffers (row-major, contiguous, f32 unless noted)
 * plus the HIP stream to launch on, and returns an int status:
 *      0            success
 *      < 0          argument validation failure (GF_E_*)
 *      > 0          hipError_t of the failing runtime call
 * The phase entry points allocate nothing, never synchronise the device and keep no state between
 * calls: everything a launch reads or writes is named by its descriptor, so they are re-entrant per
 * (device, stream) and ownership never crosses the ABI.  What IS process-global, and
 * therefore not re-entrant, is diagnostic: the tuning switches of gf_set_option() (read once per
 * launch), the single-phase event profiler gf_profile_begin() / gf_profile_end() and the append-only
 * table of run-time programs (gf_post_program_register, thread-safe).  The opaque handles of
 * gf_run_ops_graph() / gf_event_create() belong to the caller that created them.
 * No torch types appear here; the Python host (genesis_forge_amd) hands over `tensor.data_ptr()`
 * values via ctypes.
 *
 * Conventions
 *   N = num_envs, D = controlled DOFs, L = tracked links of one ContactManager,
 *   C = contact slots, T/K = number of reward/termination terms, O = observation width.
 *   Masks are 1 byte per env (torch.bool).  Counters are int32.  Quaternions are (w,x,y,z).
 *   "draws": wherever the reference consumes torch's global RNG (Tensor.uniform_), the kernel
 *   takes either a dense array of U[0,1) floats (parity mode) or NULL, in which case it
 *   generates the same kind of draw with Philox4x32-10 keyed by (seed, stream, env, lane).
 */
#ifndef GF_STEP_H
#define GF_STEP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF_ABI_VERSION 7

#define GF_MAX_TERMS 24          /* reward terms per manager          */
#define GF_MAX_TERM_TERMS 16     /* termination terms per manager     */
#define GF_MAX_OBS_ITEMS 24      /* observation items per manager     */
#define GF_MAX_CONTACT_VIEWS 4   /* ContactManagers visible to a term */
#define GF_MAX_COMMAND_VIEWS 4   /* CommandManagers visible to a term */
#define GF_MAX_EXT 16            /* opaque Python-evaluated columns   */
#define GF_MAX_LINK_IDS 32       /* target / with link ids            */
#define GF_MAX_RANGES 8          /* command ranges per CommandManager */
#define GF_MAX_OBS_WIDTH 256     /* single-frame observation width    */
#define GF_MAX_GAITS 4           /* gaits a GaitCommandManager samples from */

/* error codes */
#define GF_OK 0
#define GF_E_NULL (-1)      /* required pointer is NULL              */
#define GF_E_RANGE (-2)     /* size / count out of supported range   */
#define GF_E_OPCODE (-3)    /* unknown opcode in a term table        */
#define GF_E_SLOT (-4)      /* term references an unbound view/slot  */
#define GF_E_UNSUPPORTED (-5)

/* ------------------------------------------------------------------------------------------
 * Shared views
 * ---------------------------------------------------------------------------------------- */

/* World-frame base-link state of one entity, as returned by RigidEntity.get_pos/quat/vel/ang
 * (call sites: genesis_forge/utils.py:13-55, managers/entity_manager.py:130-146,189-195). */
typedef struct GfEntityView {
    const float* pos;      /* [N,3] */
    const float* quat;     /* [N,4] (w,x,y,z) */
    const float* lin_vel;  /* [N,3] world frame */
    const float* ang_vel;  /* [N,3] world frame */
} GfEntityView;

/* Buffers a ContactManager publishes (managers/contact/contact_manager.py:142-156,300-314). */
typedef struct GfContactView {
    const float* contacts;              /* [N,L,3] link-local net force  */
    const float* last_air_time;         /* [N,L] or NULL                 */
    const float* current_contact_time;  /* [N,L] or NULL                 */
    const float* link_vel;              /* [N,L,3] world link velocity (feet_slide, gait rewards) or NULL */
    const float* link_pos;              /* [N,L,3] world link position (foot_height_reward) or NULL */
    int32_t num_links;                  /* L */
    int32_t _pad;
} GfContactView;

/* CommandManager._command (managers/command/command_manager.py:77-81). */
typedef struct GfCommandView {
    const float* command;  /* [N,width], row n at command + n*stride */
    int32_t width;
    int32_t stride;        /* floats between rows; 0 = width (dense).  The gait manager's 14 observed columns live in 16-float rows */
} GfCommandView;

/* TerrainManager's map of the terrain (managers/terrain_manager.py:281-359): the height field after `* vertical_scale` and
 * the transpose of :354-359, i.e. rows index y and cols index x, plus the bounds get_terrain_height normalises with
 * (:117-136).  A NULL height_field is the "no height field" case of :112-114: every query returns origin_z. */
typedef struct GfTerrainView {
    const float* height_field;  /* [rows, cols] f32 metres, or NULL */
    int32_t rows;               /* H (y direction) */
    int32_t cols;               /* W (x direction) */
    float x_min;                /* (float)bounds[0] */
    float x_span;               /* (float)(bounds[1] - bounds[0]), the double difference rounded once (`div_(x_max - x_min)`) */
    float y_min;
    float y_span;
    float origin_z;
    int32_t _pad;
} GfTerrainView;

/* One row of a term table.  Meaning of p[]/i[] is per opcode (see enums below). */
typedef struct GfTerm {
    int32_t op;
    int32_t flags;
    float w;        /* rewards: (float)(weight*dt)  (reward_manager.py:185-186)  */
    float p[4];
    int32_t i[4];
    int32_t row;    /* rewards: row of episode_sums / term_out this term owns      */
} GfTerm;           /* 48 bytes */

/* Per-step scalar statistics, accumulated on device, read back lazily by the host.
 * These are the only cross-env reductions of the path; at world_size>1 the
 * summed block (cast to f64) is the payload of the single RCCL all-reduce.
 * Every `stats` pointer of this ABI addresses an array of GF_STATS_SHARDS blocks: workgroup b
 * accumulates into shard b % GF_STATS_SHARDS (1 024 waves adding to ONE word serialise at ≈ 88
 * atomics/µs; sharding removes that), and the reader sums the shards (flags: bitwise OR). */
#define GF_STATS_SHARDS 64
typedef struct GfStepStats {
    int32_t term_fired[GF_MAX_TERM_TERMS]; /* #envs for which termination term k fired (termination_manager.py:178-182) */
    int32_t reset_count;                   /* #envs reset this step (managed_env.py:308-323) */
    int32_t action_flags;                  /* bit0: NaN action seen, bit1: Inf action seen (position_action_manager.py:402-406) */
    int32_t contact_flags;                 /* bit0: non-finite contact force sanitised (contact_manager.py:399-403) */
    int32_t resample_count;                /* #envs whose command was resampled by command.step() */
    int32_t gait_count[GF_MAX_GAITS];      /* #envs currently on gait g, counted every step by gf_gait_step ("Metrics / gait_<name>_envs") */
    double reward_episode_sum[GF_MAX_TERMS]; /* Σ over reset envs of episode_sum[t]/episode_seconds (reward_manager.py:207-216) */
} GfStepStats;

/* ------------------------------------------------------------------------------------------
 * Phase A — GenesisEnv.step bookkeeping + PositionActionManager.step
 *   genesis_env.py:181-205; managers/action/base.py:67-82;
 *   managers/action/position_action_manager.py:376-419; position_within_limits.py:100-131
 * ---------------------------------------------------------------------------------------- */
enum { GF_ACTION_POSITION = 0, GF_ACTION_WITHIN_LIMITS = 1 };

typedef struct GfActionArgs {
    int32_t num_envs;
    int32_t num_dofs;
    int32_t mode;            /* GF_ACTION_* */
    int32_t check_finite;    /* !quiet_action_errors: set stats->action_flags bits */
    const float* actions_in; /* [N,D] policy output (never written) */
    const float* scale;      /* [D] */
    const float* offset;     /* [D] */
    const float* clip_lo;    /* [D] */
    const float* clip_hi;    /* [D] */
    float* env_actions;      /* [N,D] GenesisEnv._actions      (may be NULL: bookkeeping skipped) */
    float* env_last_actions; /* [N,D] GenesisEnv._last_actions */
    int32_t* episode_length; /* [N] += 1 (may be NULL) */
    float* targets;          /* [N,D] out: clamped PD targets == action_manager.get_actions() */
    GfStepStats* stats;      /* may be NULL */
    GfStepStats* stats_zero; /* optional: GF_STATS_SHARDS blocks this launch zeroes — the NEXT step's slot of a statistics
                                ring, so a recorded step needs neither a memset nor a per-step device→host copy */
    const GfStepStats* stats_fold_src; /* optional: the PREVIOUS step's slot (complete by stream order) … */
    double* stats_fold_dst;            /* … folded over its shards into this GF_STATS_VECTOR_LEN row (layout of gf_stats_pack) */
    double* stats_last_reset;          /* optional: copy of the most recent folded row whose reset_count > 0
                                          (RewardManager.last_episode_mean_reward, reward_manager.py:138-153) */
} GfActionArgs;

/* ------------------------------------------------------------------------------------------
 * Phase B2 — ContactManager.step
 *   managers/contact/kernel.py:5-90 (accumulation), contact_manager.py:384-477 (sanitise, air time)
 * ---------------------------------------------------------------------------------------- */
typedef struct GfContactArgs {
    int32_t num_envs;
    int32_t num_contacts;     /* C */
    int32_t num_scene_links;  /* second dim of links_quat */
    int32_t num_targets;      /* L */
    int32_t num_with;         /* W */
    int32_t has_with_filter;
    int32_t track_air_time;
    int32_t _pad;
    const float* force;       /* [N,C,3] world frame, force on link_b */
    const float* position;    /* [N,C,3] */
    const int32_t* link_a;    /* [N,C] */
    const int32_t* link_b;    /* [N,C] */
    const float* links_quat;  /* [N,num_scene_links,4] */
    const float* links_vel;   /* [N,num_scene_links,3] world link velocities, or NULL */
    float* link_vel_out;      /* [N,L,3] out: velocity of each tracked link (what rewards.feet_slide reads, rewards.py:496-504), or NULL */
    const float* links_pos;   /* [N,num_scene_links,3] world link positions, or NULL */
    float* link_pos_out;      /* [N,L,3] out: position of each tracked link (gait foot_height_reward), or NULL */
    int32_t target_link_ids[GF_MAX_LINK_IDS];
    int32_t with_link_ids[GF_MAX_LINK_IDS];
    float air_time_threshold; /* (float)air_time_contact_threshold */
    float dt;                 /* (float)scene.dt */
    float* contacts;          /* [N,L,3] out */
    float* contact_positions; /* [N,L,3] out (mean position) */
    float* position_counts;   /* [N,L]   out */
    float* last_air_time;     /* [N,L] in/out, NULL unless track_air_time */
    float* current_air_time;
    float* last_contact_time;
    float* current_contact_time;
    GfStepStats* stats;       /* may be NULL */
} GfContactArgs;

/* ------------------------------------------------------------------------------------------
 * Phase B3 — TerminationManager.step   (managers/termination_manager.py:151-190,
 *                                        mdp/terminations.py)
 * ---------------------------------------------------------------------------------------- */
enum {
    GF_T_TIMEOUT = 1,              /* episode_length > max_episode_length          (terminations.py:17-23)  */
    GF_T_BAD_ORIENTATION = 2,      /* p0 = tilt threshold in sin-space, i0 = grace  (terminations.py:26-71)  */
    GF_T_BASE_HEIGHT_BELOW = 3,    /* p0 = minimum height                           (terminations.py:74-99)  */
    GF_T_OUT_OF_BOUNDS = 4,        /* p0..p3 = xmin,xmax,ymin,ymax incl. margin     (terminations.py:102-137)*/
    GF_T_HAS_CONTACT = 5,          /* i0 = contact view, p0 = thr, i1 = min_contacts (terminations.py:139-155)*/
    GF_T_CONTACT_FORCE = 6,        /* i0 = contact view, p0 = thr                   (terminations.py:158-172)*/
    GF_T_CONTACT_FORCE_GRACE = 7,  /* i0 = view, p0 = thr, i1 = grace steps         (terminations.py:175-205)*/
    GF_T_EXTERNAL = 8              /* i0 = ext slot (bool column evaluated by the host) */
};
#define GF_SPAWN_BLOCK 0x100000u  /* Philox counter block of the spawn draws: (x, y, rot z, -) and, in the next block, (rot x, rot y, -, -) */
#define GF_TERM_FLAG_TIME_OUT 1    /* TerminationConfigItem.time_out → OR into truncated */

typedef struct GfTerminationArgs {
    int32_t num_envs;
    int32_t num_terms;
    GfEntityView entity;
    const int32_t* episode_length;
    const int32_t* max_episode_length;  /* may be NULL → timeout never fires */
    GfContactView contact[GF_MAX_CONTACT_VIEWS];
    const uint8_t* ext[GF_MAX_EXT];
    uint8_t* terminated;   /* [N] out */
    uint8_t* truncated;    /* [N] out */
    uint8_t* term_out;     /* [K,N] optional: raw per-term masks (direct mdp.* calls) */
    GfStepStats* stats;    /* may be NULL */
    GfTerm terms[GF_MAX_TERM_TERMS];
} GfTerminationArgs;

/* ------------------------------------------------------------------------------------------
 * Phase B4 — RewardManager.step   (managers/reward_manager.py:166-195, mdp/rewards.py)
 * ---------------------------------------------------------------------------------------- */
enum {
    GF_R_IS_ALIVE = 1,           /* rewards.py:31-37  */
    GF_R_TERMINATED = 2,         /* rewards.py:40-46  */
    GF_R_BASE_HEIGHT = 3,        /* p0 = target | i0 = command view (flag CMD); flag TERRAIN: minus the terrain height under the base  rewards.py:54-90 */
    GF_R_DOF_SIMILAR_TO_DEFAULT = 4, /* rewards.py:93-109  */
    GF_R_LIN_VEL_Z_L2 = 5,       /* rewards.py:112-135 */
    GF_R_ANG_VEL_XY_L2 = 6,      /* rewards.py:138-161 */
    GF_R_FLAT_ORIENTATION_L2 = 7,/* rewards.py:164-193 */
    GF_R_BODY_ACCEL_EXP = 8,     /* p0 = sensitivity, i0 = state slot, flag FIRST_CALL  rewards.py:196-249 */
    GF_R_ACTION_RATE_L2 = 9,     /* rewards.py:257-271 */
    GF_R_CMD_TRACK_LIN_VEL = 10, /* i0 = command view, p0 = sensitivity   rewards.py:279-317 */
    GF_R_CMD_TRACK_ANG_VEL = 11, /* i0 = command view, i1 = column, p0 = sensitivity  rewards.py:320-358 */
    GF_R_STAND_STILL = 12,       /* i0 = command view, p0 = command threshold  rewards.py:361-385 */
    GF_R_HAS_CONTACT = 13,       /* i0 = contact view, p0 = thr, i1 = min_contacts  rewards.py:393-410 */
    GF_R_CONTACT_FORCE = 14,     /* i0 = contact view, p0 = thr   rewards.py:413-428 */
    GF_R_FEET_AIR_TIME = 15,     /* i0 = contact view, i1 = command view or -1, p0 = time_threshold, p1 = max-thr (flag MAX), p2 = (float)(dt+1e-8)  rewards.py:431-469 */
    GF_R_FEET_SLIDE = 16,        /* i0 = contact view  rewards.py:472-504 */
    GF_R_EXTERNAL = 17,          /* i0 = ext slot ([N] f32 column evaluated by the host) */
    /* GaitCommandManager rewards (examples/gait_trainer/gait_command_manager.py:278-345): i0 = contact view of the feet
     * (contacts, link_vel, link_pos), i1 = command view of the gait state rows (GF_GAIT_* columns), i2 = the row of each foot
     * (FL, FR, RL, RR) inside that contact view, one byte each */
    GF_R_GAIT_PHASE = 18,        /* exp(Σ_feet swing ? -|F| : stance ? -|v| : 0)                 :295-345
                                  * Reference quirk reproduced when GfRewardArgs.gait_wave_flags is set: the reference builds its index
                                  * lists with `mask.nonzero().flatten()` on an [N,1] mask (:335-338), which interleaves the COLUMN
                                  * indices (all 0) with the row indices — so env 0 gets the stance weights of a foot whenever ANY
                                  * env is in stance for it, else the swing weights whenever any env is in swing. */
    GF_R_FOOT_HEIGHT = 19        /* exp(-Σ_feet |v_xy| (z - foot_height)^2 / p0), p0 = sensitivity  :278-293 */
};
#define GF_RW_FLAG_CMD 1         /* base_height: target from command view i0 column 0 */
#define GF_RW_FLAG_TERRAIN 2     /* base_height: subtract get_terrain_height(pos.x, pos.y) sampled from GfRewardArgs.terrain */
#define GF_RW_FLAG_MAX 4         /* feet_air_time: clamp max */
#define GF_RW_FLAG_FIRST_CALL 8  /* body_acceleration_exp: no prev state yet */

enum { GF_REWARD_MODE_STEP = 0, GF_REWARD_MODE_EVAL = 1 };

typedef struct GfRewardArgs {
    int32_t num_envs;
    int32_t num_dofs;
    int32_t num_terms;
    int32_t mode;              /* GF_REWARD_MODE_* */
    float dt;                  /* (float)env.dt, added to episode_seconds */
    int32_t logging_enabled;
    GfEntityView entity;
    const float* dof_pos;          /* [N,D] */
    const float* default_dof_pos;  /* [D]   */
    const float* actions;          /* [N,D] raw env.actions      */
    const float* last_actions;     /* [N,D] raw env.last_actions */
    const uint8_t* terminated;     /* [N] extras["terminations"] */
    GfContactView contact[GF_MAX_CONTACT_VIEWS];
    GfCommandView command[GF_MAX_COMMAND_VIEWS];
    const float* ext[GF_MAX_EXT];
    float* state[4];           /* body_acceleration_exp prev (lin,ang) body-frame velocities, [N,6] each */
    float* reward;             /* [N]   out (STEP) */
    float* episode_sums;       /* [T,N] in/out (STEP, logging) */
    float* episode_seconds;    /* [N]   in/out (STEP) */
    float* term_out;           /* [T,N] out (EVAL): unweighted term values */
    GfTerrainView terrain;     /* base_height(terrain_manager=…): sampled in the kernel (rewards.py:84-88) */
    const uint8_t* gait_wave_flags; /* GfGaitArgs.wave_flags of the gait manager GF_R_GAIT_PHASE reads (env-0 quirk), or NULL */
    GfTerm terms[GF_MAX_TERMS];
} GfRewardArgs;

/* ------------------------------------------------------------------------------------------
 * Phase B5 — CommandManager.step / reset / resample_command
 *   managers/command/command_manager.py:152-170,290-303
 * ---------------------------------------------------------------------------------------- */
enum { GF_CMD_STEP = 0, GF_CMD_MASKED = 1, GF_CMD_ALL = 2 };

typedef struct GfCommandArgs {
    int32_t num_envs;
    int32_t num_ranges;       /* R */
    int32_t mode;             /* GF_CMD_* */
    int32_t resample_steps;   /* int(resample_time_sec/dt) */
    const int32_t* episode_length; /* STEP */
    const uint8_t* mask;      /* MASKED: resample where mask!=0 */
    const uint8_t* mask2;     /* MASKED: optional second mask OR-ed in (terminated|truncated) */
    const float* draws;       /* [N,R] U[0,1) or NULL → Philox */
    uint64_t seed;
    uint64_t stream;          /* distinct per (manager, call) so draws never repeat */
    uint32_t env_offset;      /* global index of local env 0 (env sharding): Philox is keyed by the GLOBAL env id */
    uint32_t _pad2;
    float lo[GF_MAX_RANGES];
    float hi[GF_MAX_RANGES];
    float* command;           /* [N,R] in/out */
    GfStepStats* stats;       /* may be NULL (STEP mode counts resamples) */
} GfCommandArgs;

/* ------------------------------------------------------------------------------------------
 * Phase B5' — GaitCommandManager.step / reset / resample_command
 *   examples/gait_trainer/gait_command_manager.py:185-255,347-399 (a user-level CommandManager in the reference's
 *   gait_trainer example).  State of one env = one 64-byte row:
 *     [0..3] foot_offset (FL,FR,RL,RR)  [4] foot_height  [5] gait_period  [6..13] clock_input (4 sin, 4 cos)
 *     [14] gait_time  [15] gait_phase
 *   so columns 0..13 are exactly `observation()` (:257-268) and 0..5 are `command` (:127-141).
 *   STEP  : resample where episode_length % resample_steps == 0 (command_manager.py:152-162), count envs per gait
 *           (_log_metrics, :430-441), then advance the clock for EVERY env (:231-239):
 *             gait_time = fmod(gait_time + dt, period); gait_phase = gait_time / period;
 *             clock[i] = sin(2π·fmod(phase + offset_i, 1)), clock[4+i] = cos(…)
 *   MASKED / ALL : resample the masked envs, zero their clock_input / gait_time / gait_phase (:241-255).
 *   Resampling (:185-211,347-399): gait g from the categorical distribution `cum_weight` (inverse CDF of draw 0; with one
 *   gait no draw is consumed), foot offsets from the gait table, foot_height = clearance_lo for the fixed-clearance gaits
 *   else U(clearance) from draw 1, gait_period = U(period) from draw 2.
 * ---------------------------------------------------------------------------------------- */
enum { GF_GAIT_OFFSET = 0, GF_GAIT_HEIGHT = 4, GF_GAIT_PERIOD = 5, GF_GAIT_CLOCK = 6, GF_GAIT_TIME = 14, GF_GAIT_PHASE = 15,
       GF_GAIT_ROW = 16, GF_GAIT_OBS_WIDTH = 14 };

typedef struct GfGaitArgs {
    int32_t num_envs;
    int32_t mode;               /* GF_CMD_STEP / GF_CMD_MASKED / GF_CMD_ALL */
    int32_t resample_steps;
    int32_t num_gaits;          /* gaits currently sampled from (curriculum), 1..GF_MAX_GAITS */
    const int32_t* episode_length; /* STEP */
    const uint8_t* mask;        /* MASKED */
    const uint8_t* mask2;       /* MASKED, optional */
    const float* draws;         /* [N,3] U[0,1) (gait select, foot clearance, gait period) or NULL → Philox */
    uint64_t seed;
    uint64_t stream;
    uint32_t env_offset;
    int32_t fixed_clearance_mask; /* bit g: gait g always uses clearance_lo ("pronk", "bound"; :366-368) */
    float cum_weight[GF_MAX_GAITS];   /* f32 cumulative sum of the normalised gait weights (:379-399) */
    float gait_offsets[GF_MAX_GAITS][4];
    float clearance_lo, clearance_hi; /* _foot_clearance_range */
    float period_lo, period_hi;       /* _gait_period_range */
    float dt;                   /* (float)env.dt */
    float two_pi;               /* (float)(2*pi) */
    float* state;               /* [N,GF_GAIT_ROW] in/out */
    int64_t* selected;          /* [N] in/out: _gait_selected */
    uint8_t* wave_flags;        /* [ceil(N/64) rounded up to 4] out, optional: for each block of 64 envs, bit 2f = some env of the
                                   block has foot f in swing, bit 2f+1 = in stance, for the CURRENT state (phi = fmod(phase +
                                   offset_f, 1)·2π; swing: 0 <= phi < π, stance: π <= phi < 2π).  Every launch that changes an env
                                   rewrites its block's byte from the 64 rows it holds: no atomics, nothing incremental.  The
                                   host initialises it for the all-zero state (every foot in swing: 0x55). */
    GfStepStats* stats;         /* STEP: gait_count[] += envs per gait; may be NULL */
} GfGaitArgs;

/* ------------------------------------------------------------------------------------------
 * Phase R — masked reset of all manager-owned state (ManagedEnvironment.reset fan-out)
 *   managed_env.py:336-371; genesis_env.py:207-254; reward_manager.py:197-222;
 *   contact_manager.py:316-329; position_action_manager.py:421-464; mdp/reset.py:67-124
 * The reference compacts done envs with nonzero() and gathers/scatters; this is the same
 * update expressed as a mask so no host sync is needed.
 * ---------------------------------------------------------------------------------------- */
typedef struct GfResetArgs {
    int32_t num_envs;
    int32_t num_dofs;
    int32_t num_reward_terms;   /* rows of episode_sums */
    int32_t num_contact;        /* contact managers with air-time state */
    const uint8_t* mask;        /* [N] reset where mask|mask2 != 0 */
    const uint8_t* mask2;       /* optional */
    /* GenesisEnv.reset */
    float* env_actions;         /* [N,D] → 0 */
    float* env_last_actions;    /* [N,D] → 0 */
    int32_t* episode_length;    /* [N]   → 0 */
    int32_t* max_episode_length;/* [N]   → round(base + (2u-1)*max_random_scaling) when scaling>0 */
    int32_t base_max_episode_length;
    float max_random_scaling;   /* (float)(base*max_episode_random_scaling); 0 disables */
    const float* len_draws;     /* [N] U[0,1) or NULL → Philox */
    /* RewardManager.reset */
    float* episode_sums;        /* [T,N] */
    float* episode_seconds;     /* [N] → 1e-10 */
    uint32_t reward_log_mask;   /* bit t set: term t has weight != 0 → value/=secs, accumulate mean */
    int32_t reward_logging;     /* logging_enabled */
    /* ContactManager.reset: 4 air-time arrays per manager */
    float* air_state[GF_MAX_CONTACT_VIEWS][4];
    int32_t air_links[GF_MAX_CONTACT_VIEWS];
    /* Scene-side state, only when the scene exposes masked setters (synthetic scene).
     * NULL pointers skip the section (real Genesis: host calls the envs_idx setters). */
    float* scene_dof_pos;       /* [N,D] ← default_dof_pos + (2u-1)*noise_scale */
    float* scene_dof_vel;       /* [N,D] ← 0 */
    const float* default_dof_pos; /* [D] */
    float dof_noise_scale;
    const float* dof_draws;     /* [N,D] U[0,1) or NULL → Philox (only read when noise_scale != 0) */
    float* scene_pos;           /* [N,3] ← reset_pos    (mdp.reset.position) */
    float* scene_quat;          /* [N,4] ← reset_quat   */
    float* quat_stash;          /* [N,4] optional: receives the pre-reset quat of every reset env (see GfObservationArgs.stale_quat) */
    float* scene_lin_vel;       /* [N,3] ← 0 when zero_velocity */
    float* scene_ang_vel;       /* [N,3] ← 0 when zero_velocity */
    float reset_pos[3];
    float reset_quat[4];
    int32_t set_quat;
    int32_t zero_velocity;
    uint64_t seed;
    uint64_t stream;
    uint32_t env_offset;        /* global index of local env 0 */
    /* mdp.reset.randomize_terrain_position (mdp/reset.py:127-226 → terrain_manager.py:170-279): with spawn_mode = 1 the base of a
     * reset env goes to x = u0*spawn_x_span + spawn_x_min, y = u1*spawn_y_span + spawn_y_min (the usable centre area),
     * z = terrain height there + spawn_height_offset, instead of reset_pos; with spawn_set_quat = 1 its orientation becomes
     * xyz_to_quat(rx, ry, rz) where axis k is U(spawn_rot_lo[k], spawn_rot_hi[k]) if bit k of spawn_rot_mask is set and 0
     * otherwise (define_quat only writes the axes given as (lo, hi) tuples, reset.py:172-196; default {"z": (0, 2π)}). */
    int32_t spawn_mode;
    int32_t spawn_set_quat;
    int32_t spawn_rot_mask;
    float spawn_x_min, spawn_x_span;
    float spawn_y_min, spawn_y_span;
    float spawn_height_offset;
    float spawn_rot_lo[3], spawn_rot_hi[3];
    const float* spawn_draws;   /* [N,5] U[0,1) (x, y, rot x, rot y, rot z) or NULL → Philox blocks GF_SPAWN_BLOCK, +1 of `stream` */
    GfTerrainView terrain;
    GfStepStats* stats;         /* may be NULL */
} GfResetArgs;

/* ------------------------------------------------------------------------------------------
 * Phase O — ObservationManager.get_observations
 *   managers/observation_manager.py:218-256, mdp/observations.py
 * ---------------------------------------------------------------------------------------- */
enum {
    GF_O_COMMAND = 1,          /* i0 = command view; width = view width */
    GF_O_ANG_VEL_BODY = 2,     /* EntityManager.get_angular_velocity  (entity_manager.py:142-146) */
    GF_O_LIN_VEL_BODY = 3,     /* EntityManager.get_linear_velocity   (entity_manager.py:136-140) */
    GF_O_PROJ_GRAVITY = 4,     /* EntityManager.get_projected_gravity (entity_manager.py:130-134) */
    GF_O_DOF_POS = 5,          /* action_manager.get_dofs_position()  */
    GF_O_DOF_VEL = 6,          /* action_manager.get_dofs_velocity()  */
    GF_O_DOF_FORCE = 7,        /* action_manager.get_dofs_force()     */
    GF_O_ACTIONS = 8,          /* action_manager.get_actions() == clamped targets (quirk q1) */
    GF_O_RAW_ACTIONS = 9,      /* env.actions */
    GF_O_CONTACT_FORCE_NORM = 10, /* i0 = contact view; width = L  (observations.py:182-193) */
    GF_O_EXTERNAL = 11,        /* i0 = ext slot, i1 = width: [N,width] f32 evaluated by the host */
    GF_O_BASE_POS = 12,
    GF_O_BASE_QUAT = 13
};

typedef struct GfObsItem {
    int32_t op;
    int32_t width;
    int32_t i0;
    int32_t i1;
    float scale;   /* ObservationConfigItem.scale (1.0 = none) */
    float noise;   /* item noise or manager noise; 0 = none     */
} GfObsItem;

typedef struct GfObservationArgs {
    int32_t num_envs;
    int32_t num_dofs;
    int32_t num_items;
    int32_t obs_width;        /* O = Σ width */
    int32_t history_len;      /* H >= 1 */
    int32_t history_ring;     /* 0: `obs` = [new frame | the first H-1 frames of prev_obs] (newest first, observation_manager.py:218-226:
                                 the previous OUTPUT is the history, so a launch moves (2H-1)·O floats per env);
                                 k+1: IN-PLACE RING — `obs` is the persistent [N, H, O] history itself and the launch writes ONLY the new
                                 frame, into frame slot k (0 <= k < H); prev_obs is unused.  O floats per env instead of (2H-1)·O; the
                                 caller walks the slots from k upwards (wrapping) to read newest first: the host passes
                                 k = (H - step % H) % H so that this walk is ascending in memory. */
    GfEntityView entity;
    const float* dof_pos;     /* [N,D] */
    const float* dof_vel;     /* [N,D] */
    const float* dof_force;   /* [N,D] */
    const float* targets;     /* [N,D] */
    const float* env_actions; /* [N,D] */
    GfContactView contact[GF_MAX_CONTACT_VIEWS];
    GfCommandView command[GF_MAX_COMMAND_VIEWS];
    const float* ext[GF_MAX_EXT];
    const float* noise_draws; /* [N,O] U[0,1) or NULL → Philox */
    uint64_t seed;
    uint64_t stream;
    uint32_t env_offset;      /* global index of local env 0 */
    uint32_t ring_slots;      /* history_ring != 0 only.  0: the ring has history_len frame slots (`obs` = [N, H, O]).  S >= history_len: `obs` is
                                 [N, S, O] — the env stride is S·O floats — and history_ring - 1 < S.  With more slots than frames the host can
                                 hand out H consecutive slots as a newest-first WINDOW of the buffer (a strided view, no gather): it walks the
                                 slot downwards and mirrors the newest H-1 frames behind slot S' when the walk wraps
                                 (ObservationManager output="window").  (This word was padding before: 0 = the old behaviour.) */
    /* The reference's EntityManager caches base_quat at entity.step() and does not refresh it after the reset
     * that follows in the same tick (entity_manager.py:163-167,189-195): body-frame items of envs that were just
     * reset are rotated by their PRE-reset quaternion.  stale_quat (written by gf_masked_reset.quat_stash) supplies
     * it for envs whose stale_mask|stale_mask2 byte is set; NULL disables. */
    const float* stale_quat;  /* [N,4] */
    const uint8_t* stale_mask;
    const uint8_t* stale_mask2;
    const float* prev_obs;    /* [N,O*H] previous output (history shift source); NULL when H==1 */
    float* obs;               /* [N,O*H] out, newest frame first (observation_manager.py:224-226) */
    GfObsItem items[GF_MAX_OBS_ITEMS];
} GfObservationArgs;

/* ------------------------------------------------------------------------------------------
 * Entity helpers — EntityManager.get_projected_gravity / get_linear_velocity /
 * get_angular_velocity as standalone calls (entity_manager.py:130-146; utils.py:13-55).
 * ---------------------------------------------------------------------------------------- */
enum { GF_ROT_PROJ_GRAVITY = 0, GF_ROT_LIN_VEL = 1, GF_ROT_ANG_VEL = 2 };
typedef struct GfRotateArgs {
    int32_t num_envs;
    int32_t what;         /* GF_ROT_* */
    GfEntityView entity;
    float* out;           /* [N,3] */
} GfRotateArgs;

/* ------------------------------------------------------------------------------------------
 * TerrainManager.get_terrain_height (managers/terrain_manager.py:100-166): bilinear sample of the shared
 * height field at world (x, y), `F.grid_sample(mode="bilinear", padding_mode="border", align_corners=True)`:
 *   nx = ((x - x_min) / x_span) * 2 - 1            (the in-place chain of :123-136, one rounding per op)
 *   ix = clamp(((nx + 1) / 2) * (W - 1), 0, W - 1) (unnormalise + border clip), same for iy with H
 *   out = nw*v[y0,x0] + ne*v[y0,x1] + sw*v[y1,x0] + se*v[y1,x1], weights (x1-ix)(y1-iy) …, taps outside the field
 *   contribute nothing (only x1 = W / y1 = H, where the weight is 0 anyway).
 * x and y are strided so `pos[:, 0]`, `pos[:, 1]` of an [N,3] tensor (rewards.py:85-87) or columns of the spawn
 * buffer (terrain_manager.py:236-239) are read in place.
 * ---------------------------------------------------------------------------------------- */
typedef struct GfTerrainHeightArgs {
    int64_t num;           /* number of queries */
    const float* x;        /* element i at x[i * x_stride] */
    const float* y;
    int64_t x_stride;      /* in floats */
    int64_t y_stride;
    GfTerrainView terrain;
    float* out;            /* [num] */
} GfTerrainHeightArgs;

/* ------------------------------------------------------------------------------------------
 * Synthetic scene tick — stands in for Genesis' scene.step() in benchmarks and parity tests
 * (SURVEY.md §7 step 5).  Deterministic, integer-Philox driven, f32 ops in a fixed order so the
 * HIP kernel, the C oracle and the numpy model used to drive the reference agree bit for bit.
 * ---------------------------------------------------------------------------------------- */
typedef struct GfSynthSceneArgs {
    int32_t num_envs;
    int32_t num_dofs;
    int32_t num_contacts;     /* C (0: no contact generation) */
    int32_t num_scene_links;
    float dt;
    float joint_rate;         /* first-order tracking gain [1/s] */
    float ang_noise;          /* rad/s per tick */
    float lin_noise;          /* m/s per tick */
    float height_target;
    float contact_prob;       /* per slot (walking model: per BODY slot) */
    float contact_force;      /* force scale */
    float foot_contact_prob;  /* walking model (foot_link_mask != 0): mean per-tick probability that a foot touches the ground */
    const float* targets;     /* [N,D] PD targets written by phase A */
    float* pos;               /* [N,3] in/out */
    float* quat;              /* [N,4] in/out */
    float* lin_vel;           /* [N,3] in/out */
    float* ang_vel;           /* [N,3] in/out */
    float* dof_pos;           /* [N,D] in/out */
    float* dof_vel;           /* [N,D] out    */
    float* contact_force_out; /* [N,C,3] or NULL */
    float* contact_pos_out;   /* [N,C,3] */
    int32_t* link_a_out;      /* [N,C] */
    int32_t* link_b_out;      /* [N,C] */
    float* links_quat_out;    /* [N,num_scene_links,4] */
    float* links_vel_out;     /* [N,num_scene_links,3] or NULL */
    float* links_pos_out;     /* [N,num_scene_links,3] or NULL: world position of every link (RigidEntity.get_links_pos) */
    uint64_t seed;
    uint64_t tick;
    uint32_t env_offset;      /* global index of local env 0 */
    /* Walking contact model.  0: every slot is an independent (ground, random robot link) contact with probability contact_prob.
     * Otherwise bit l marks scene link l (< 32) as a FOOT: the k-th foot owns slot k — its contact with the ground, made with
     * probability min(1, 1.8 p) while the foot is in the stance half of a 20-tick trot cycle (diagonal pairs alternate, the
     * cycle offset differs per env) and 0.2 p in the swing half, p = foot_contact_prob (its normal force comes from the draw that picks the side): two to four feet of a quadruped on the
     * ground every tick, as the reference's foot ContactManager / feet_air_time / the Taichi kernel's matching branch expect
     * (examples/gait_trainer/environment.py:140-153, mdp/rewards.py:431-469, managers/contact/kernel.py:47-78).  The remaining
     * slots are body contacts as before (probability contact_prob, any robot link).  Every contact puts the ground on side a or
     * side b at random (force stored as the force on link_b, so it is negated when the robot link is link_a): both branches of
     * kernel.py:74-78 run. */
    uint32_t foot_link_mask;
} GfSynthSceneArgs;

/* ------------------------------------------------------------------------------------------
 * Rollout storage write (first slice): what the RL library does with a step's outputs.
 * rsl_rl's OnPolicyRunner (call site examples/simple/train.py:125-129, `num_steps_per_env` = 24, :75) copies every
 * step's observation, reward and done flags into a time-major RolloutStorage with three `copy_` launches
 * (`observations[step].copy_(obs)`, `rewards[step].copy_(rew)`, `dones[step].copy_(dones)`).  Here the step's own
 * kernel stores them: the fused post-physics launch writes its observation tile, reward and masks a second time,
 * straight into the storage rows (GfPostRefs.rollout), or gf_rollout_write does it as one launch.
 * Layout (plain pointers, time-major, contiguous): observations [T+1, N, W] f32, rewards [T, N] f32, dones [T, N] u8.
 * Step t of a rollout writes rewards[t], dones[t] = terminated | truncated, and observations[t+1] = the observation the
 * step RETURNS (the policy's next input; row T is the bootstrap observation, row 0 the one the rollout started from).
 * The caller passes the row addresses, so the library needs neither T nor t.
 * ---------------------------------------------------------------------------------------- */
typedef struct GfRolloutArgs {
    int32_t num_envs;
    int32_t obs_width;          /* W: floats per env of `obs` (O·H of the policy ObservationManager) */
    const float* obs;           /* [N, W] the observation this step returns */
    const float* reward;        /* [N] */
    const uint8_t* terminated;  /* [N] */
    const uint8_t* truncated;   /* [N] */
    float* obs_out;             /* &observations[t+1][0][0], or NULL */
    float* reward_out;          /* &rewards[t][0], or NULL */
    uint8_t* done_out;          /* &dones[t][0], or NULL */
} GfRolloutArgs;

/* ------------------------------------------------------------------------------------------
 * The policy's half of a transition and the return computation (remainder).  rsl_rl's OnPolicyRunner
 * (third-party; configured and called by the reference at examples/simple/train.py:37-79,125-129: PPO, gamma 0.99, lam 0.95,
 * num_steps_per_env 24) stores, besides the env's outputs, what the policy produced for the step — actions, value estimate,
 * log-probability, action mean and std: five more `copy_` launches in RolloutStorage.add_transitions — after bootstrapping
 * time-outs into the reward (PPO.process_env_step: rewards += gamma * values * time_outs), and at the end of a rollout runs
 * RolloutStorage.compute_returns: a loop over the T steps, backwards, of eight elementwise launches each (GAE, Schulman et
 * al. 2016), then normalises the advantages — ~200 launches for T = 24.
 *   gf_rollout_policy_write: the five rows and the time-out bootstrap in ONE launch.
 *   gf_gae:  for each env (one lane; rows of the time-major arrays are read and written coalesced), t = T-1 … 0:
 *       next  = t == T-1 ? last_values[n] : values[t+1][n];   live = 1 - dones[t][n]
 *       delta = rewards[t][n] + live * gamma * next - values[t][n]
 *       adv   = delta + live * gamma * lam * adv;            returns[t][n] = adv + values[t][n]
 *     advantages[t][n] = returns[t][n] - values[t][n]; with `normalize`, a second launch maps them to
 *     (adv - mean) / (std + 1e-8) over all T*N entries (unbiased std, as torch.std), from f64 moments the first launch
 *     accumulates (one atomic pair per wave).  One f32 operation per step of the recurrence in this order: bit-identical
 *     to the torch loop before normalisation.  Algorithmic traffic: R 9 + W 8 bytes per (step, env) (+ R/W 8 to normalise).
 * ---------------------------------------------------------------------------------------- */
typedef struct GfRolloutPolicyArgs {
    int32_t num_envs;
    int32_t num_actions;        /* A */
    const float* actions;       /* [N, A] sampled actions, or NULL (each source may be NULL together with its row) */
    const float* values;        /* [N] value estimates */
    const float* log_prob;      /* [N] */
    const float* mu;            /* [N, A] action mean */
    const float* sigma;         /* [N, A] action std */
    float* actions_out;         /* &actions[t][0][0] of the time-major storage */
    float* values_out;          /* &values[t][0] */
    float* log_prob_out;        /* &actions_log_prob[t][0] */
    float* mu_out;              /* &mu[t][0][0] */
    float* sigma_out;           /* &sigma[t][0][0] */
    const uint8_t* time_outs;   /* [N] truncated flags of this step, or NULL: reward_row[n] += gamma * values[n] * time_outs[n] */
    float* reward_row;          /* &rewards[t][0] (read-modify-write), required with time_outs */
    float gamma;
    int32_t _pad;
} GfRolloutPolicyArgs;

typedef struct GfGaeArgs {
    int32_t num_envs;
    int32_t num_steps;          /* T >= 1 */
    const float* rewards;       /* [T, N] */
    const float* values;        /* [T, N] */
    const uint8_t* dones;       /* [T, N] */
    const float* last_values;   /* [N] value of the bootstrap observation */
    float gamma, lam;
    float* returns;             /* [T, N] */
    float* advantages;          /* [T, N] */
    double* moments;            /* [2] scratch: sum and sum of squares of the advantages; required with normalize */
    int32_t normalize;          /* 1: advantages <- (advantages - mean) / (std + 1e-8) */
    int32_t _pad;
} GfGaeArgs;

/* ------------------------------------------------------------------------------------------
 * The index list of a step's done envs.  The reference compacts `(terminated | truncated).nonzero()` on every step
 * (managed_env.py:308-310) and hands the list to `reset(ids)`; on the masked path nothing needs it — except code that takes
 * index lists by contract: Genesis' `envs_idx` setters (position_action_manager.py:455-464, mdp/reset.py:102-124), a user's
 * `reset(ids)` override, user-defined manager classes.  torch's `nonzero()` there is an OR launch, a two-pass select, a
 * device-to-host copy of the count and an allocation.  gf_done_compact: one small launch up to 131 072 envs (a block counts the masks in front of it itself), two above — per-block counts, then
 * offsets + ordered writes — produce the ascending list in a caller-owned buffer and the count in a word the host reads after
 * synchronising the stream (pinned host memory).  Same order as nonzero(), so everything downstream is unchanged.
 * ---------------------------------------------------------------------------------------- */
typedef struct GfCompactArgs {
    int64_t num_envs;
    const uint8_t* mask;      /* [N] nonzero = listed */
    const uint8_t* mask2;     /* [N] OR-ed in, or NULL */
    int64_t* ids_out;         /* [N] capacity; the first *count_out entries are the indices, ascending */
    int32_t* count_out;       /* one word: device memory or pinned host memory */
    int32_t* block_counts;    /* scratch, ceil(N / 4096) + 1 words */
    int32_t wait;             /* 1: the call returns after the stream has drained — *count_out (pinned host memory) is then valid
                               * for the caller to read: the one synchronisation of a step that needs the list, taken inside the
                               * call instead of through a second trip into the runtime */
    int32_t _pad;
} GfCompactArgs;

/* ------------------------------------------------------------------------------------------
 * History ring -> the reference's observation layout.  The reference keeps a list of H frames, pops the oldest, inserts the new
 * one in front and returns `torch.cat(self._history, dim=-1)` (observation_manager.py:219-226): every call writes a NEW
 * [N, H*O] tensor, newest frame first.  With the history kept as an in-place ring (GfObservationArgs.history_ring: the step
 * writes O floats per env) that returned tensor is one streaming gather: out[n, j*O + c] = ring[n, (k + j) mod H, c] with k the
 * slot of the newest frame.  Unlike shifting the previous output inside the step's kernel, the copy has no dependency on the
 * step's arithmetic and runs as thousands of short workgroups at the copy rate of the memory system; it is also the caller's
 * private tensor, so the default output mode needs no clone on top.
 * ---------------------------------------------------------------------------------------- */
typedef struct GfHistoryUnrollArgs {
    const float* ring;        /* [N, H, O]: the `obs` of launches with history_ring != 0 */
    float* out;               /* [N, H*O], 16-byte aligned */
    float* out2;              /* optional second destination with the same layout (a rollout-storage row), or NULL */
    int64_t num_envs;
    int32_t frame_width;      /* O >= 1 */
    int32_t history_len;      /* H >= 1 */
    int32_t ring_slot;        /* the history_ring value of the launch that wrote the newest frame: k + 1 */
    int32_t _pad;
} GfHistoryUnrollArgs;

/* ------------------------------------------------------------------------------------------
 * Entry points.  `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream).
 * ---------------------------------------------------------------------------------------- */
int gf_abi_version(void);
/* library-wide tuning switches (process global).  GF_OPT_POST_VARIANT selects the fused post-physics kernel:
 * 0 = table interpreter, one wave per 64-env tile; 1 = table interpreter, four specialised waves per tile;
 * 2 = (default) as 1, plus the static programs: a config whose structure matches a registered program runs
 * straight-line code compiled for it.  All variants produce bit-identical results. */
/* GF_OPT_PROFILE_STRIDE: gf_profile_begin stamps every k-th launch of the profiled phase (default 1 = every launch); a
 * stamped launch costs the host several microseconds more than a plain one, so a timed region samples instead. */
/* GF_OPT_GRAPH (default 0): 1 = gf_run_ops_graph replays a recorded step as one hipGraphLaunch.  Off by default because it
 * MEASURED slower on MI355X / ROCm 7.2 (every node's arguments change every step, so each replay pays one
 * hipGraphExecKernelNodeSetParams per kernel on top of the graph launch): Go2 command config 21.7 vs 18.1 µs/step at 4 096
 * envs, 27.8 vs 24 µs at 65 536; gait config 107 vs 94 µs at 8 192 (profiles/r01_l_graph_vs_plain.jsonl). */
/* GF_OPT_CHAIN (default 1): gf_run_ops folds runs of per-env phases the fused post-physics kernel does not cover into phase
 * chains — termination → reward → command.step…, and reset → command.reset… → observe… — one launch each (csrc/gf_chain.hip). */
/* GF_OPT_FOLD_CONTACT (default 1): gf_run_ops hands the contact_step ops in front of a fused post-physics op to that launch
 * (gf_post_physics_step_contacts) instead of launching the contact kernel itself — unless that measured slower (more than 4 tracked
 * links below 16 384 envs, more than 12 below 32 768); 0 keeps the two launches (A/B, tests), 2 folds whenever it is possible. */
enum { GF_OPT_POST_VARIANT = 0, GF_OPT_PROFILE_STRIDE = 1, GF_OPT_GRAPH = 2, GF_OPT_CHAIN = 3, GF_OPT_FOLD_CONTACT = 4, GF_OPT_COUNT = 5 };
int gf_set_option(int option, int value);
int gf_sizeof(int which);   /* sizeof of the ABI structs (0 = GfStepStats … 11 = GfObsItem, 12 GfTerrainView, 13 GfTerrainHeightArgs, 14 GfGaitArgs, 15 GfContactView, 16 GfCommandView, 17 GfPostRefs, 18 GfRolloutArgs, 19 GfHistoryUnrollArgs, 20 GfRolloutPolicyArgs, 21 GfGaeArgs, 22 GfCompactArgs): binding self-check */
const char* gf_build_info(void);
const char* gf_error_string(int code);

int gf_stats_clear(GfStepStats* stats, void* stream);            /* zero all GF_STATS_SHARDS blocks */

int gf_action_step(const GfActionArgs* a, void* stream);          /* replaces genesis_env.py:181-205 + position_action_manager.py:376-419 */
int gf_contact_step(const GfContactArgs* a, void* stream);        /* replaces contact_manager.py:331-336 (+ contact/kernel.py:5-90) */
int gf_termination_step(const GfTerminationArgs* a, void* stream);/* replaces termination_manager.py:151-190 */
int gf_reward_step(const GfRewardArgs* a, void* stream);          /* replaces reward_manager.py:166-195 */
int gf_command_step(const GfCommandArgs* a, void* stream);        /* replaces command_manager.py:152-170,290-303 */
int gf_gait_step(const GfGaitArgs* a, void* stream);              /* replaces examples/gait_trainer/gait_command_manager.py:185-255 */
int gf_masked_reset(const GfResetArgs* a, void* stream);          /* replaces managed_env.py:336-366 fan-out */
int gf_observe(const GfObservationArgs* a, void* stream);         /* replaces observation_manager.py:218-256 */
int gf_entity_rotate(const GfRotateArgs* a, void* stream);        /* replaces entity_manager.py:130-146 */
int gf_terrain_height(const GfTerrainHeightArgs* a, void* stream);/* replaces terrain_manager.py:100-166 */
int gf_synth_scene_step(const GfSynthSceneArgs* a, void* stream); /* stands in for scene.step() (managed_env.py:292) */
int gf_rollout_write(const GfRolloutArgs* a, void* stream);       /* replaces the RolloutStorage copy_ launches of the RL library (examples/simple/train.py:125-129) */
int gf_history_unroll(const GfHistoryUnrollArgs* a, void* stream);/* replaces the torch.cat of observation_manager.py:226 */
int gf_rollout_policy_write(const GfRolloutPolicyArgs* a, void* stream);   /* the policy's rows of a transition + time-out bootstrap (rsl_rl add_transitions; call site examples/simple/train.py:125-129) */
int gf_done_compact(const GfCompactArgs* a, void* stream);       /* replaces the nonzero() of managed_env.py:308-310 where an index list is still needed */
int gf_gae(const GfGaeArgs* a, void* stream);                     /* returns and advantages of a finished rollout (rsl_rl compute_returns; gamma / lam: examples/simple/train.py:41-47) */

/* ------------------------------------------------------------------------------------------
 * Fused post-physics step: everything ManagedEnvironment.step() does after scene.step() and the
 * contact managers — termination → reward → command.step → reset of done envs → command.reset →
 * observations (managed_env.py:303-326) — as ONE launch.  All of it is per-env work on the same
 * state (pos/quat/vel/ang, the [N,D] rows, commands), so one lane carries an env through every phase
 * with that state in registers: 566 B/env of traffic instead of 268 + 26 + 4 + 388 + the reset's
 * re-reads, one launch instead of six.  The descriptors are the per-phase ones (so the semantics are by
 * definition those of calling the phases in sequence — which is what the oracle twin does); the call
 * packs them into one kernarg block.  Returns GF_E_UNSUPPORTED when the combination cannot be fused
 * (host-evaluated columns without GF_POST_TERMINATION_DONE — see below —, host-evaluated observation items, parity-mode draws,
 * more than 32 DOF, phases reading different buffers, more than 2 command / 1 gait /
 * 2 observation managers …): the caller then runs the phases one by one (or leaves a manager out of the call and runs it behind).
 * ---------------------------------------------------------------------------------------- */
#define GF_POST_MAX_CMD 2
#define GF_POST_MAX_OBS 2
#define GF_POST_MAX_GAIT 1

/* GfPostRefs.flags.  GF_POST_TERMINATION_DONE: the termination phase of this step has ALREADY run as a launch of its own
 * (gf_termination_step on `termination`: masks and statistics are final) — the fused launch reads `terminated` / `truncated`
 * instead of evaluating the terms.  This is how a step whose task config carries Python-level terms still fuses everything behind
 * them: a host-evaluated reward column (GF_R_EXTERNAL) has to be computed after the termination phase and before the reward
 * phase (managed_env.py:303-319: the callable may read this step's termination buffers), so the step is
 * termination launch -> the callables -> ONE launch for reward ... observation.  Any termination table is then acceptable,
 * including GF_T_EXTERNAL terms.
 * GF_POST_OBSERVE_ONLY: every phase of the step up to and including the reset has ALREADY run — the termination phase, the
 * rewards, the command managers and the reset of the done envs (gf_masked_reset on `reset`, by mask or by the index list a user's
 * `reset(envs_idx)` override took: managed_env.py:308-310) — and only the observation managers are left (managed_env.py:324).
 * The launch reads `terminated` / `truncated` (an env reset in this tick is observed through `reset->quat_stash`, the stale-cache
 * quirk of entity_manager.py:189-195) and runs the observation waves of the fused kernel for up to two managers as one launch;
 * `reward`, the command and gait lists must be empty.  This is the observation part of a step whose reset goes through user
 * code.
 * GF_POST_NO_RESET: the counterpart for the FRONT of such a step — termination, rewards, command.step / gait.step as one launch,
 * and nothing after them: no env is reset (the masks are written; the reset follows by index list through user code, then the
 * observation-only launch), `reset`, `command_reset[]`, `gait_reset[]` may be NULL, `num_observe` must be 0. */
enum { GF_POST_TERMINATION_DONE = 1, GF_POST_OBSERVE_ONLY = 2, GF_POST_NO_RESET = 4 };
typedef struct GfPostRefs {
    const GfTerminationArgs* termination;                 /* required */
    const GfRewardArgs* reward;                           /* may be NULL */
    const GfResetArgs* reset;                             /* required; mask/mask2 must be termination's outputs */
    int32_t num_command;
    int32_t num_observe;
    const GfCommandArgs* command_step[GF_POST_MAX_CMD];   /* mode GF_CMD_STEP  */
    const GfCommandArgs* command_reset[GF_POST_MAX_CMD];  /* mode GF_CMD_MASKED on the same masks, same buffers */
    const GfObservationArgs* observe[GF_POST_MAX_OBS];
    /* GaitCommandManagers stepped / reset inside the same launch (examples/gait_trainer: BASELINE config 5).  The swing / stance
     * bytes (GfGaitArgs.wave_flags) are read ACROSS 64-env blocks by GF_R_GAIT_PHASE (env-0 quirk) while every block rewrites its
     * own byte, so a single launch needs two buffers: it reads gait_step->wave_flags (the state the previous step left; must be
     * the reward descriptor's gait_wave_flags) and writes every block's byte for the state it leaves into gait_flags_next, which
     * the caller makes the current buffer afterwards (ping-pong).  Sequentially the result is the same: gait.step rewrites every
     * block's byte, gait.reset the blocks with a reset env, both from the rows they hold. */
    int32_t num_gait;
    int32_t flags;                                    /* GF_POST_* bits */
    const GfGaitArgs* gait_step[GF_POST_MAX_GAIT];    /* mode GF_CMD_STEP */
    const GfGaitArgs* gait_reset[GF_POST_MAX_GAIT];   /* mode GF_CMD_MASKED on the termination masks, same state */
    uint8_t* gait_flags_next[GF_POST_MAX_GAIT];       /* same size as wave_flags; may be NULL when wave_flags is NULL */
    /* optional: the launch also stores the step's observation / reward / done flags into rollout-storage rows (§8f-5).
     * rollout->obs must be the obs buffer of one of `observe`, ->reward the reward buffer, the masks termination's outputs. */
    const GfRolloutArgs* rollout;
} GfPostRefs;

int gf_post_physics_check(const GfPostRefs* r);               /* GF_OK if gf_post_physics_step can fuse this combination */
int gf_post_physics_step(const GfPostRefs* r, void* stream);  /* replaces managed_env.py:303-326 in one launch */
/* The same launch with the scene's ContactManagers stepped IN FRONT of the other phases (SURVEY.md §8f-1 ∘ §8f-2): the reference runs
 * contact.step, termination, reward … back to back per env (managed_env.py:294-326, managers/contact/contact_manager.py:384-477,
 * managers/contact/kernel.py:35-90), and the fused launch works on the same 64-env tiles and reads what the contact step just
 * wrote — so `contacts[0 .. num_contacts)` (up to four GfContactArgs over the same scene arrays: what gf_contact_step takes, one
 * per manager) run as the first phase of the launch: slot ids staged in the tile's LDS, one lane per (env, tracked link), the
 * managers' public buffers written as gf_contact_step writes them, a workgroup barrier, then termination … observation.  By
 * construction the results are those of gf_contact_step per manager followed by gf_post_physics_step.  GF_E_UNSUPPORTED when the
 * managers cannot be folded (more than 32 tracked links or 16 with-filter links, link ids above 254, slot rows beyond the tile's
 * LDS, a statistics block other than the step's, GF_POST_OBSERVE_ONLY): the caller runs gf_contact_step itself. */
int gf_post_physics_step_contacts(const GfPostRefs* r, const GfContactArgs* const* contacts, int num_contacts, void* stream);
/* Writes "program <id> (<name>): <structure signature>" for this combination into buf: which kernel
 * gf_post_physics_step would launch (id 0 = table interpreter, >0 = a static program compiled for exactly this
 * structure) and the signature in the notation of csrc/gf_post_programs.h.  Host-only, no GPU needed. */
int gf_post_physics_describe(const GfPostRefs* r, char* buf, int cap);
/* Static programs for configs the library was not built with.  The reference's configs are live Python dicts
 * (managers/config/config_item.py:31-44; reward_manager.py:166-195 walks whatever the dict holds), so a user's task is not
 * one of the compiled-in structures; it would run the table interpreter (about 1.3 x the kernel time).  Instead the host
 * generates the program struct of the recorded step's signature (gf_post_physics_describe), compiles csrc/gf_post_ws.h with it
 * for gfx950 into a small shared object (hipcc, a few seconds, cached by signature; genesis_forge_amd/_programs.py) and
 * registers it here: from then on gf_post_physics_step launches the plugin's post_ws_kernel<Program> for every descriptor
 * that matches it — ids from 100 upwards in gf_post_physics_describe.  Structure only: weights, parameters, ranges, scales
 * stay run-time kernel arguments, a config whose STRUCTURE changes simply stops matching (interpreter until its own program is
 * there).  The plugin exports gfp_abi_version / gfp_args_size (checked against this library: GF_E_UNSUPPORTED on a mismatch or
 * an unloadable file) / gfp_name / gfp_matches / gfp_kernel / gfp_lds_bytes.  Append-only, at most 64, thread-safe. */
int gf_post_program_register(const char* plugin_path, int* program_id_out);
int gf_post_program_count(void);

/* ------------------------------------------------------------------------------------------
 * Recorded step: the fixed launch sequence of one ManagedEnvironment.step() replayed with a single
 * call.  The host records the (phase, descriptor) pairs of one ordinary step, keeps the descriptors
 * alive, patches the few per-step fields (action pointer, RNG stream ids, observation ring slots) in
 * place and replays — the phase order of managed_env.py:274-334 without per-launch host work.
 * ---------------------------------------------------------------------------------------- */
typedef struct GfOp {
    int32_t phase;      /* GF_PHASE_* or GF_OP_* */
    int32_t _pad;
    const void* args;   /* the phase's descriptor (GfActionArgs*, …) */
} GfOp;

enum { GF_OP_STATS_CLEAR = 100, GF_OP_STATS_COPY = 101, GF_OP_POST_PHYSICS = 102 /* args = GfPostRefs* */,
       GF_OP_STATS_PACK = 103 /* args = GfStatsPackArgs* */ };

/* Fold the GF_STATS_SHARDS shards of one statistics block into the f64 vector that is all-reduced across ranks
 * (layout: term_fired[GF_MAX_TERM_TERMS], reset_count, nan_flag, inf_flag, contact_flag, resample_count,
 * reward_episode_sum[GF_MAX_TERMS], gait_count[GF_MAX_GAITS]; flags fold with max, everything else adds). */
#define GF_STATS_VECTOR_LEN (GF_MAX_TERM_TERMS + 5 + GF_MAX_TERMS + GF_MAX_GAITS)
typedef struct GfStatsPackArgs {
    const GfStepStats* src;   /* device, GF_STATS_SHARDS blocks */
    double* dst;              /* device, GF_STATS_VECTOR_LEN */
} GfStatsPackArgs;
int gf_stats_pack(const GfStatsPackArgs* a, void* stream);

/* rows: [num_rows <= 64][GF_STATS_VECTOR_LEN] consecutive packed (and, with a process group, all-reduced) statistics rows, oldest
 * first.  Copies the NEWEST row whose reset_count entry is > 0 to dst; leaves dst alone when no row reset anything.  This keeps
 * "the episode means of the last reset" (reward_manager.py:138-153, :197-222 — the reference refreshes them inside every reset())
 * on the device when ring rows are recycled unread, so the host never has to read a row just to carry that value forward. */
int gf_stats_last_reset(const double* rows, int num_rows, double* dst, void* stream);

typedef struct GfStatsCopyArgs {
    const GfStepStats* src;   /* device, GF_STATS_SHARDS blocks */
    void* dst;                /* pinned host, GF_STATS_SHARDS * sizeof(GfStepStats) */
    void* event;              /* from gf_event_create(); recorded after the copy; may be NULL */
} GfStatsCopyArgs;

int gf_run_ops(const GfOp* ops, int num_ops, void* stream, int* failed_index);

/* Recorded step, patched natively: the few descriptor fields that change from one step to the next are described ONCE as a
 * patch table (which host address receives what), and a step is a single call — apply the table, then gf_run_ops — instead of
 * a Python closure per phase (managed_env.py:274-334 has no counterpart: the reference re-marshals every op every step).
 * All state lives in caller-owned memory (the descriptors, the counters and rotors the table points at): the call itself is
 * stateless and re-entrant per recorded step. */
enum {
    GF_PATCH_ACTIONS = 1,    /* *(const void**)target = actions                                    (the policy output of this step) */
    GF_PATCH_STREAM = 2,     /* *(uint64_t*)target = ++*rng_stream                                 (Philox stream ids, in draw order) */
    GF_PATCH_COUNTER = 3,    /* *(uint64_t*)target = (*(uint64_t*)aux)++                           (scene tick) */
    GF_PATCH_ROTATE = 4,     /* r = (GfRotor*)aux: if target: *(void**)target = r->slot[r->cur]; r->cur = (r->cur + 1) % r->count;
                                if target2: *(void**)target2 = r->slot[r->cur]                     (output slots, ping-pong buffers) */
    GF_PATCH_PARAM = 5,      /* *(const void**)target = params[index]                              (statistics ring slots) */
    GF_PATCH_COPY = 6,       /* *(uint64_t*)target = *(const uint64_t*)aux                         (a field that follows another) */
    GF_PATCH_RING_SLOT = 7,  /* c = (GfRingClock*)aux: *(int32_t*)target = (c->length - c->calls % c->length) % c->length + 1; ++c->calls;
                                if target2: *(int32_t*)target2 = the same value
                                                              (GfObservationArgs.history_ring, GfHistoryUnrollArgs.ring_slot) */
    GF_PATCH_PARAM_OFFSET = 8 /* *(const char**)target = (const char*)params[index] + (intptr_t)aux
                                 (a scene with Genesis' public surface only: every getter — entity.get_pos() …,
                                 collider.get_contacts(), rigid_solver.get_links_quat(); entity_manager.py:189-195,
                                 contact_manager.py:384-400 — returns a NEW tensor each tick; params[index] is this tick's tensor,
                                 aux the byte offset of the field's view inside it.  A NULL params[index] is refused: GF_E_NULL) */
};
typedef struct GfRotor {
    int32_t cur;
    int32_t count;           /* 1..8 */
    void* slot[8];
} GfRotor;
typedef struct GfRingClock {
    int32_t calls;
    int32_t length;
} GfRingClock;
typedef struct GfReplayPatch {
    int32_t kind;            /* GF_PATCH_* */
    int32_t index;           /* GF_PATCH_PARAM */
    void* target;
    void* target2;
    void* aux;
} GfReplayPatch;
typedef struct GfReplay {
    const GfOp* ops;         /* may be NULL / num_ops 0: patch only (the caller then replays the ops in pieces) */
    int32_t num_ops;
    int32_t num_patches;
    const GfReplayPatch* patches;
    uint64_t* rng_stream;    /* the env's stream counter (GF_PATCH_STREAM pre-increments it) */
} GfReplay;
int gf_replay_step(const GfReplay* r, const void* actions, const void* const* params, int num_params, void* stream, int* failed_index);
/* The same replay as ONE hipGraphLaunch: the first call builds a linear hipGraph of the step's kernel launches, later calls
 * refresh every node's kernel arguments in place (the descriptors change from step to step: action pointer, RNG streams,
 * ring slots) and launch the graph — the host pays one graph launch instead of one launch per kernel.  *cache is an opaque
 * handle (start with NULL, release with gf_graph_destroy).  Falls back to gf_run_ops when the step contains memset / copy
 * ops, a phase is being profiled, GF_OPT_GRAPH is 0 or the graph API refuses. */
int gf_run_ops_graph(void** cache, const GfOp* ops, int num_ops, void* stream, int* failed_index);
int gf_graph_destroy(void** cache);
void* gf_event_create(void);
int gf_event_destroy(void* event);
int gf_event_synchronize(void* event);   /* blocks the host until the event has completed */

/* Optional per-phase HIP-event timing used by bench.py (events recorded on `stream`
 * immediately around the kernel launch of the selected phase). */
enum { GF_PHASE_ACTION = 0, GF_PHASE_CONTACT, GF_PHASE_TERMINATION, GF_PHASE_REWARD, GF_PHASE_COMMAND,
       GF_PHASE_RESET, GF_PHASE_OBSERVE, GF_PHASE_ROTATE, GF_PHASE_SCENE, GF_PHASE_POST, GF_PHASE_TERRAIN, GF_PHASE_GAIT, GF_PHASE_ROLLOUT,
       GF_PHASE_UNROLL, GF_PHASE_ROLLOUT_POLICY, GF_PHASE_GAE, GF_PHASE_COMPACT, GF_PHASE_COUNT };
int gf_profile_begin(int phase, int max_samples);     /* start recording event pairs for `phase` */
int gf_profile_end(double* total_ms, int* samples);    /* sync events, return Σ elapsed + count, free them */

#ifdef __cplusplus
}
#endif
#endif /* GF_STEP_H */
