// gf_post_args.h — packed descriptor of the fused post-physics launch and the small device helpers its kernels share
// (gf_post.hip: single-wave interpreter; gf_post_ws.h: wave-specialised interpreter and the static programs).
#pragma once

#include "gf_launch.h"
#include "gf_terms.h"
#include "gf_contact_tile.h"

// Diagnostic build only (tools/stamp_post.hip, -DGF_STAMPS): lane 0 of one workgroup records the 100 MHz wall
// clock at the phase boundaries into a buffer of its own; no product build contains a stamp.
#ifdef GF_STAMPS
extern "C" unsigned long long* gf_debug_stamps;  // host variable set by the tool
#define GF_STAMP(i)                                                                                          \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (a.stamps && blockIdx.x == a.stamp_block && threadIdx.x == 0) {                                   \
            a.stamps[i] = __builtin_amdgcn_s_memrealtime();                                                  \
            a.stamps[16 + i] = __builtin_amdgcn_s_memtime();                                                 \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
// per-wave stamps of the wave-specialised kernel: stamps[64 + 16*wave + i]
#define GF_WSTAMP(i)                                                                                         \
    do {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        if (a.stamps && blockIdx.x == a.stamp_block && (threadIdx.x & 63) == 0)                              \
            a.stamps[64 + 16 * wave + i] = __builtin_amdgcn_s_memrealtime();                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    } while (0)
#else
#define GF_STAMP(i)
#define GF_WSTAMP(i)
#endif

namespace gf {

constexpr int kPostMaxTerm = 8;
constexpr int kPostMaxReward = 16;
constexpr int kPostAuxRows = 32;  // per-lane scratch rows: 4 per float4 chunk of a DOF row, up to D = 28
constexpr int kPostMaxItems = 12;
constexpr int kPostMaxRanges = 4;

struct PostCmd {
    float* command;
    int32_t width;
    int32_t resample_steps;
    uint64_t stream_step;
    uint64_t stream_reset;
    float lo[kPostMaxRanges];
    float hi[kPostMaxRanges];
};

// one GaitCommandManager stepped and reset inside the launch (gf_gait.hip's gait_body, both modes, on wave 0's registers)
constexpr int kViewGait = GF_POST_MAX_CMD;   // cmd_of_view value: the view aliases the gait manager's state rows
struct PostGait {
    float* state;               // [N, GF_GAIT_ROW]
    int64_t* selected;          // [N]
    const uint8_t* flags_in;    // swing / stance bytes the previous step left (read by GF_R_GAIT_PHASE for env 0), or NULL
    uint8_t* flags_out;         // every block's byte for the state this launch leaves (the caller swaps the two), or NULL
    uint64_t stream_step, stream_reset;
    int32_t resample_steps, num_gaits, fixed_clearance_mask, _pad;
    float cum_weight[GF_MAX_GAITS];
    float gait_offsets[GF_MAX_GAITS][4];
    float clearance_lo, clearance_hi, period_lo, period_hi, dt, two_pi;
};

struct PostObs {
    float* obs;
    const float* prev;
    uint64_t stream;
    int32_t num_items;
    int32_t width;
    int32_t history;
    int32_t ring;      // GfObservationArgs.history_ring: 0 = shift, k+1 = in-place ring, the new frame goes to slot k
    int32_t ring_slots;   // GfObservationArgs.ring_slots: 0 = the ring has `history` slots; S = `obs` is [N, S, O] (a window buffer)
    int32_t _pad;
    GfObsItem items[kPostMaxItems];
};

// An observation output of this size or more is streamed out with the non-temporal hint: nothing in the step reads it again, and the
// policy that does finds 200 MB at 1 M envs in no cache either way; cached, it evicts the state the next launch reads (whole step at
// 1 048 576 envs 229 -> 212 us with the hint; at 65 536 envs, 12.6 MB, no difference — profiles/r03_z_ab_nt.jsonl).
constexpr int64_t kObsStreamBytes = (int64_t)64 << 20;

// The scene's ContactManagers stepped IN FRONT of the other phases by the same launch (SURVEY.md §8f-1 ∘ §8f-2;
// managed_env.py:294-326 runs contact.step, termination, reward … back to back per env, contact_manager.py:384-477): a compact
// image of up to four GfContactArgs over the same scene arrays.  num_mgr == 0: the contact phase ran as a launch of its own (or the
// config has no ContactManager).  Link ids travel as bytes (a scene with more than 255 links keeps the stand-alone launch).
constexpr int kFoldMaxMgr = 4;
constexpr int kFoldMaxTargets = 32;   // tracked links over all folded managers
constexpr int kFoldMaxWith = 16;      // with-filter links per manager
struct PostContactMgr {
    float *contacts, *contact_positions, *position_counts, *link_vel_out, *link_pos_out;
    float *last_air_time, *current_air_time, *last_contact_time, *current_contact_time;
    float air_time_threshold;
    uint8_t num_targets, num_with, has_with_filter, track_air_time;
    uint8_t with_ids[kFoldMaxWith];
};
static_assert(sizeof(PostContactMgr) == 96 && offsetof(PostContactMgr, air_time_threshold) == 72, "the kernel reads this image word by word");
struct PostContact {
    const float *force, *position, *links_quat, *links_vel, *links_pos;
    const int32_t *link_a, *link_b;
    int32_t num_contacts, num_scene_links, num_mgr, total_targets;
    float dt;
    int32_t _pad;
    uint8_t target_ids[kFoldMaxTargets], mgr_of[kFoldMaxTargets], local_of[kFoldMaxTargets];
    PostContactMgr m[kFoldMaxMgr];
};
// LDS of the contact phase, in 32-bit words: the tables the lanes index (manager images, tracked-link table, with-filter lists),
// then the tile's slot ids and occupancy masks (gf_contact_tile.h).  It aliases the LDS of the phases behind it.
constexpr int kFoldTableWords = kFoldMaxMgr * kContactMgrWords + kFoldMaxTargets + kFoldMaxTargets / 2 + kFoldMaxMgr * GF_MAX_LINK_IDS;
static_assert(kFoldTableWords % 4 == 0, "the slot-id rows behind the tables are read and written as 16-byte units");
// (+ one scratch row per wave: the target of the contact phase's cache warm-up requests)
__host__ __device__ constexpr size_t fold_lds_bytes(int C) { return (size_t)(kFoldTableWords + contact_lds_ints(kEnvBlock, C) + 4 * kEnvBlock) * 4; }

struct alignas(16) GfPostArgs {
    int32_t num_envs, num_dofs, num_term, num_rew;
    uint32_t needs;
    int32_t n_cmd, n_obs, logging;
    float dt;
    int32_t reward_rows;
    uint32_t reward_log_mask;
    uint32_t uncovered_rows;   // rows of episode_sums no active term owns (zero weight): still zeroed on reset
    uint64_t seed;
    uint32_t env_offset;
    int32_t has_maxlen;
    // state
    float *pos, *quat, *lin_vel, *ang_vel;       // entity views (writable: scene-side reset)
    float *dof_pos, *dof_vel;
    const float *dof_force, *targets, *default_dof_pos;
    float *env_actions, *env_last_actions;
    int32_t *episode_length, *max_episode_length;
    uint8_t *terminated, *truncated;
    float *reward, *episode_sums, *episode_seconds;
    GfStepStats* stats;
    float* quat_stash;
    // views shared by every phase (slot indices in the copied terms/items are remapped onto these)
    GfContactView contact[GF_MAX_CONTACT_VIEWS];
    GfCommandView command[GF_MAX_COMMAND_VIEWS];
    int32_t cmd_of_view[GF_MAX_COMMAND_VIEWS];   // index into cmds[] of the manager that owns the view's buffer, or -1
    const float* ext[GF_MAX_EXT];   // host-evaluated reward columns (GF_R_EXTERNAL)
    float* state[4];
    // reset
    int32_t scene_reset, set_quat, zero_velocity, reset_env /* bit0: actions rows, bit1: episode_length */, reset_dofs;
    int32_t base_max_episode_length;
    float max_random_scaling, dof_noise_scale;
    float reset_pos[3], reset_quat[4];
    uint64_t stream_reset;
    // mdp.reset.randomize_terrain_position (field names as in GfResetArgs so spawn_pose() serves both) + the terrain map the
    // spawn and base_height(terrain_manager=…) sample
    int32_t spawn_mode, spawn_set_quat, spawn_rot_mask;
    float spawn_x_min, spawn_x_span, spawn_y_min, spawn_y_span, spawn_height_offset;
    float spawn_rot_lo[3], spawn_rot_hi[3];
    GfTerrainView terrain;
    float* air_state[GF_MAX_CONTACT_VIEWS][4];
    int32_t air_links[GF_MAX_CONTACT_VIEWS];
    int32_t n_air;
    int32_t n_gait;
    // rollout-storage rows of the RL library (GfRolloutArgs, §8f-5): the observation of manager `roll_obs_index`, the reward and
    // terminated | truncated are stored a second time, there
    float* roll_obs;
    float* roll_reward;
    uint8_t* roll_done;
    int32_t roll_obs_index;
    int32_t term_done;   // GF_POST_TERMINATION_DONE: the masks are inputs, the termination table is not evaluated
    int32_t obs_only;    // GF_POST_OBSERVE_ONLY: the masks are inputs and nothing is reset; the observation waves run (interpreter only)
    int32_t no_reset;    // GF_POST_NO_RESET: termination … command / gait step only: no env is treated as done, nothing is observed
    uint32_t obs_stream;   // bit m: observation manager m's output is larger than the caches keep from step to step: non-temporal stores
    const uint8_t* gait_wave_flags;   // == gait.flags_in when the reward terms reproduce the env-0 quirk (GF_R_GAIT_PHASE)
    PostGait gait;
    GfTerm tterms[kPostMaxTerm];
    GfTerm rterms[kPostMaxReward];
    PostCmd cmds[GF_POST_MAX_CMD];
    PostObs obs[GF_POST_MAX_OBS];
    PostContact cfold;
#ifdef GF_STAMPS
    unsigned long long* stamps;
    uint32_t stamp_block;
#endif
};
static_assert(sizeof(GfPostArgs) <= 4096, "GfPostArgs must fit the 4 KB kernarg segment");

enum : uint32_t {
    PN_POS = 1, PN_QUAT = 2, PN_LIN = 4, PN_ANG = 8, PN_DOFPOS = 16, PN_DOFVEL = 32, PN_TARGETS = 64, PN_ACTIONS = 128, PN_LAST = 256,
    PN_EPLEN = 512, PN_MAXLEN = 1024, PN_DOFDEV = 2048, PN_ACTRATE = 4096, PN_DOFFORCE = 8192,
};

template <> struct HasGaitTerms<GfPostArgs> { static constexpr bool value = true; };

// scale / noise of one observation element (observation_manager.py:242-250); everything it needs arrives by value
struct ObsFin {
    float scale;
    bool scaled;
};
__device__ __forceinline__ float obs_finish(const ObsFin& f, float v, int) { return f.scaled ? v * f.scale : v; }

// All ≤ 4 range draws of one command resample come out of ONE Philox block (columns 0..3 share counter col>>2 == 0),
// exactly the values philox_uniform(seed, stream, env, j) returns for j = 0..3.
// (register-only signature: a real call, no stack, so the rarely-taken resample paths cost one Philox body in the binary)
__device__ __noinline__ float4 draw_unit4(uint64_t seed, uint64_t stream, uint32_t genv, uint32_t block) {
    const U4 r = philox4x32_10(genv, block, (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    return make_float4(u24_to_unit(r.x), u24_to_unit(r.y), u24_to_unit(r.z), u24_to_unit(r.w));
}

// ---- [N, D] rows as DV = ceil(D / 4) float4 chunks of registers ------------------------------------------------------------------
// D % 4 == 0 (every static program; D is then a compile-time 4·DV and the tail code folds away): a row is DV aligned 16-byte accesses.
// Otherwise rows are only dword aligned (the same 16-byte instructions take them) and the LAST chunk holds D - 4·(DV-1) < 4 floats of
// the row: it is read and written element by element (a 16-byte access would touch the next env's row — or, for the last env, memory
// behind the array) and its missing floats read as zero, so sums over a row need no mask.
typedef float f32x4d __attribute__((ext_vector_type(4), aligned(4)));
template <int DV>
__device__ __forceinline__ void row_load(float4 (&r)[DV], const GF_GLOBAL float* p, const int D) {
#pragma unroll
    for (int c = 0; c < DV - 1; ++c) {
        const f32x4d v = *reinterpret_cast<const GF_GLOBAL f32x4d*>(p + 4 * c);
        r[c] = make_float4(v.x, v.y, v.z, v.w);
    }
    const GF_GLOBAL float* q = p + 4 * (DV - 1);
    const int t = D - 4 * (DV - 1);
    if (t >= 4) {
        const f32x4d v = *reinterpret_cast<const GF_GLOBAL f32x4d*>(q);
        r[DV - 1] = make_float4(v.x, v.y, v.z, v.w);
    } else {   // in-bounds, unconditional element loads (clamped index), then selects
        const float x = q[0], y = q[t > 1 ? 1 : 0], z = q[t > 2 ? 2 : 0];
        r[DV - 1] = make_float4(x, t > 1 ? y : 0.f, t > 2 ? z : 0.f, 0.f);
    }
}
template <int DV>
__device__ __forceinline__ void row_store(GF_GLOBAL float* p, const float4 (&r)[DV], const int D) {
#pragma unroll
    for (int c = 0; c < DV - 1; ++c) *reinterpret_cast<GF_GLOBAL f32x4d*>(p + 4 * c) = f32x4d{r[c].x, r[c].y, r[c].z, r[c].w};
    GF_GLOBAL float* q = p + 4 * (DV - 1);
    const int t = D - 4 * (DV - 1);
    const float4 l = r[DV - 1];
    if (t >= 4) {
        *reinterpret_cast<GF_GLOBAL f32x4d*>(q) = f32x4d{l.x, l.y, l.z, l.w};
    } else {
        q[0] = l.x;
        if (t > 1) q[1] = l.y;
        if (t > 2) q[2] = l.z;
    }
}

// one [D] row of registers → the lane's observation tile row (separate call per source keeps every index static)
template <int DV>
__device__ __forceinline__ void put_row(const ObsFin& f, const float4 (&r)[DV], float* row, int col, const int D) {
#pragma unroll
    for (int c = 0; c < DV; ++c) {
        const bool last = c == DV - 1;
        const int t = last ? D - 4 * (DV - 1) : 4;   // floats of this chunk that belong to the row
        row[col + 4 * c + 0] = obs_finish(f, r[c].x, col + 4 * c + 0);
        if (t > 1) row[col + 4 * c + 1] = obs_finish(f, r[c].y, col + 4 * c + 1);
        if (t > 2) row[col + 4 * c + 2] = obs_finish(f, r[c].z, col + 4 * c + 2);
        if (t > 3) row[col + 4 * c + 3] = obs_finish(f, r[c].w, col + 4 * c + 3);
    }
}

// wave-uniform value that lives in a VGPR (read from the LDS-staged descriptor) → SGPR
template <typename T>
__device__ __forceinline__ T uni(T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "uni: 4- or 8-byte types");
    if constexpr (sizeof(T) == 4) {
        uint32_t u = __builtin_bit_cast(uint32_t, v);
        u = __builtin_amdgcn_readfirstlane(u);
        return __builtin_bit_cast(T, u);
    } else {
        uint64_t u = __builtin_bit_cast(uint64_t, v);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
        u = ((uint64_t)hi << 32) | lo;
        return __builtin_bit_cast(T, u);
    }
}

}  // namespace gf
