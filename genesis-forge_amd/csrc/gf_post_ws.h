// gf_post_ws.h — wave-specialised fused post-physics kernel (included by gf_post.hip).
//
// In-kernel stamps showed the single-wave post_kernel to be ISSUE-bound: a CDNA SIMD hands a wave one instruction
// every 4 cycles, so ≈ 5 500 instructions of one stream are ≈ 9 µs however little memory they move, and at N = 65 536
// there are only 1 024 waves for 1 024 SIMDs.  Two answers live here, in ONE kernel source:
//
//  1. Wave specialisation.  A workgroup of 4 waves shares a 64-env tile (lane = env in every wave), each wave runs a
//     quarter of the step:
//       wave 0  control : base state → body-frame vectors, termination terms, command.step / command.reset draws; after the
//                         barrier: masks + env-level + base-pose reset stores
//       wave 1  reward  : [N,D] rows → Σ|dof-default|, Σ(Δaction)², episode sums (LDS-DMA); after the barrier: the term fold
//       wave 2  obs-rows: dof_pos / dof_vel rows; after the barrier: DOF reset of done envs, dof_pos / dof_vel items
//       wave 3  obs-misc: targets / raw-action rows; after the barrier: command, body-frame, action, contact items
//     then all 256 lanes stream the observation tile out.  What the waves exchange (masks, body-frame vectors, new
//     commands) goes through 5 KB of LDS and one s_barrier.  Every global load happens before the barrier and every store of
//     state that another wave loads happens after it, so the phases see exactly the values they see in post_kernel.
//
//  2. Static programs.  The kernel is templated on a policy P.  Interp<DV> walks the term / item tables at run time (they
//     are staged in LDS and every row is decoded with scalar instructions — most of the instruction count).  A static
//     program (gf_post_programs.h) fixes the STRUCTURE of a task config at compile time — which opcodes, in which order,
//     their flags and view slots, the observation layout — so the same source unrolls into straight-line code with the
//     switches folded away; weights, thresholds, ranges, pointers and sizes stay run-time values read from the kernel
//     arguments (SGPRs).  pack() selects a program only when the packed tables match its signature exactly; everything else
//     runs the interpreter.  Same statements per env in the same order ⇒ bit-identical results (tests/test_trace.py).
#pragma once

#include <utility>

#include "gf_obs_hist.h"
#include "gf_post_args.h"

namespace gf {

static_assert(kObsBlock == 4 * kEnvBlock, "the history helpers assume the 4-wave workgroup");
constexpr int kWsBlock = 4 * kEnvBlock;
enum : int { X_TERM = 0, X_TRUNC = 1, X_DONE = 2, X_BLIN = 3, X_BANG = 6, X_GRAV = 9, X_CMD = 12, X_DIRTY = 20, X_FIELDS = 21,
             X_GAIT = X_FIELDS /* the gait manager's post-step, post-reset state row: only laid out when the launch carries one */ };
__host__ __device__ constexpr int x_fields(int n_gait) { return X_FIELDS + (n_gait > 0 ? GF_GAIT_ROW : 0); }

// ---- signature rows of a static program ---------------------------------------------------------------------------------------
struct TermSig { int op, flags; };
struct RewSig { int op, flags, i0, i1; };
struct ItemSig { int op, width, i0; bool scaled, noisy; };

// optional signature member `static constexpr bool term_done` (absent = false)
template <class P, class = void>
struct prog_term_done : std::false_type {};
template <class P>
struct prog_term_done<P, std::void_t<decltype(P::term_done)>> : std::bool_constant<P::term_done> {};

template <int DV_, bool TAIL_ = false>
struct Interp {
    static constexpr bool kStatic = false;
    static constexpr int DV = DV_;
    // kTail: D % 4 != 0 — rows are only dword aligned and the last chunk is short (row_load / row_store); without it D = 4·DV is a
    // compile-time constant like in the static programs (a run-time test per row load cost the 12-DOF interpreter 8 % at 1 M envs)
    static constexpr bool kTail = TAIL_;
};

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <class P>
__device__ __forceinline__ const GfPostArgs& pick_args(const GfPostArgs& k, const float* l) {
    if constexpr (P::kStatic) return k;
    else return *reinterpret_cast<const GfPostArgs*>(l);
}
// a uniform value: already scalar when it comes from the kernel arguments, read-first-lane when it comes from the LDS copy
template <class P, class T>
__device__ __forceinline__ T uni_p(const T& v) {
    if constexpr (P::kStatic) return v;
    else return uni(v);
}
#define UNI(x) uni_p<P>(x)
#define GF_INLINE_LAMBDA __attribute__((always_inline))

template <class P>
__host__ __device__ constexpr int ws_sum_rows() {
    if constexpr (P::kStatic) return P::n_rew;
    else return kPostMaxReward;
}
template <class P>
__host__ __device__ constexpr int ws_aux_rows() { return 4 * P::DV; }
// Reward terms whose value depends on global memory only — contact forces, link velocities / positions, a gait manager's row as
// the PREVIOUS step left it — not on anything another wave of the tile computes.  A static program evaluates them BEFORE the
// barrier, in the wave with the least to do there (wave 2: three row loads), and parks the values in LDS rows: their loads go out together with every
// other load of the tile instead of as round trips of their own behind the barrier (profiles/r02_c_gait_fused_attribution.txt:
// 6.3 of the gait program's 18 µs at 8 192 envs were that chain).  The fold then reads the parked value: same function, same
// inputs, same arithmetic — bit-identical.
__host__ __device__ constexpr bool reward_op_memory_only(int op) { return reward_op_has_rows(op); }   // (gf_terms.h: term_rows / term_value)
template <class P>
__host__ __device__ constexpr int ws_pre_slot(int upto) {   // parked rows in front of term `upto` (upto = n_rew: all of them)
    if constexpr (P::kStatic) {
        int n = 0;
        for (int k = 0; k < upto && k < P::n_rew; ++k) n += reward_op_memory_only(P::rew[k].op) ? 1 : 0;
        return n;
    } else {
        return 0;
    }
}
template <class P>
__host__ __device__ constexpr int ws_term_rows() {
    if constexpr (P::kStatic) return P::n_term > 0 ? P::n_term : 1;
    else return 1;
}
template <class P>
__host__ __device__ constexpr bool ws_rew_body_frame() {   // does any reward term of the program read a body-frame vector?
    if constexpr (P::kStatic) {
        for (int k = 0; k < P::n_rew; ++k)
            if (reward_op_body_frame(P::rew[k].op)) return true;
    }
    return false;
}
template <class P>
__host__ __device__ constexpr int ws_pre_rows() {
    if constexpr (P::kStatic) return ws_pre_slot<P>(P::n_rew);
    else return 0;
}
// The same idea for OBSERVATION items that read global memory the reset does not touch: contact-force norms (parked as LDS rows by
// wave 3 before the barrier) and the dof_force row (loaded into wave 3's third, otherwise unused row register set).  Behind the
// barrier such a load was a round trip of its own — for a second manager (the gait task's critic) behind the first manager's tile
// barrier as well.
template <class P>
__host__ __device__ constexpr int ws_obs_norm_slot(int m_upto, int i_upto) {   // norm rows in front of item (m_upto, i_upto)
    if constexpr (P::kStatic) {
        int n = 0;
        for (int m = 0; m < P::n_obs && m <= m_upto; ++m)
            for (int i = 0; i < P::obs_items[m] && (m < m_upto || i < i_upto); ++i)
                n += P::item[m][i].op == GF_O_CONTACT_FORCE_NORM ? P::item[m][i].width : 0;
        return n;
    } else {
        return 0;
    }
}
template <class P>
__host__ __device__ constexpr int ws_obs_norm_rows() {
    if constexpr (P::kStatic) return ws_obs_norm_slot<P>(P::n_obs - 1, P::n_obs > 0 ? P::obs_items[P::n_obs - 1] : 0);
    else return 0;
}
template <class P>
__host__ __device__ constexpr bool ws_obs_has(int op) {
    if constexpr (P::kStatic) {
        for (int m = 0; m < P::n_obs; ++m)
            for (int i = 0; i < P::obs_items[m]; ++i)
                if (P::item[m][i].op == op) return true;
    }
    return false;
}
static_assert(kPostAuxRows >= 4 * 8, "aux rows cover 32 DOF");
// Can this kernel run the scene's ContactManagers as its first phase (GfPostArgs.cfold)?  The interpreter always can; a static program
// only when its structure reads a ContactManager's buffers — the phase costs registers (88), and a program that never sees contacts
// (the benchmark's: 79) keeps its occupancy.  A config whose contact buffers only Python reads keeps the stand-alone contact launch.
__host__ __device__ constexpr bool reward_op_reads_contacts(int op) {
    return op == GF_R_HAS_CONTACT || op == GF_R_CONTACT_FORCE || op == GF_R_FEET_AIR_TIME || op == GF_R_FEET_SLIDE || op == GF_R_GAIT_PHASE || op == GF_R_FOOT_HEIGHT;
}
template <class P>
__host__ __device__ constexpr bool ws_prog_folds() {
    if constexpr (P::kStatic) {
        if (P::n_term == 0 && P::n_rew == 0 && P::n_cmd == 0 && P::n_gait == 0) return false;   // an observation-only launch (GF_POST_OBSERVE_ONLY) never folds
        if (P::n_air > 0) return true;
        for (int k = 0; k < P::n_term; ++k)
            if (term_op_counts_contacts(P::term[k].op)) return true;
        for (int k = 0; k < P::n_rew; ++k)
            if (reward_op_reads_contacts(P::rew[k].op)) return true;
        return ws_obs_has<P>(GF_O_CONTACT_FORCE_NORM);
    } else {
        return true;
    }
}

#ifndef GF_WS_AHEAD
#define GF_WS_AHEAD 0
#endif
constexpr int kWsTilesLdsFloats = GF_WS_AHEAD > 0 ? 4 * kEnvBlock : 0;   // (experiment: one scratch row per wave, the target of the look-ahead requests)

template <class P>
__global__ __launch_bounds__(kWsBlock) void post_ws_kernel(const GfPostArgs karg) {
    constexpr int DV = P::DV;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kArgVec = P::kStatic ? 0 : (int)(sizeof(GfPostArgs) / 16);
    if constexpr (!P::kStatic) {
        const auto* src = (const __attribute__((address_space(4))) f32x4*)__builtin_amdgcn_kernarg_segment_ptr();
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < kArgVec; i += kWsBlock) dst[i] = src[i];
        __syncthreads();
    }
    const GfPostArgs& a = pick_args<P>(karg, lds);
    bool has_gait = false;
    if constexpr (P::kStatic) has_gait = P::n_gait > 0;
    else has_gait = UNI(a.n_gait) > 0;
    float* xch = lds + kArgVec * 4 + kWsTilesLdsFloats;      // [x_fields][64]
    // a static program knows how many reward rows it has and every variant knows its DOF chunks: the LDS a workgroup asks for decides
    // how many of them a CU holds (Go2 programs: 30 KB → 22.5 KB, five → seven workgroups per CU, what their 72 VGPRs allow)
    constexpr int kSumRows = ws_sum_rows<P>(), kAuxRows = ws_aux_rows<P>();
    float* lds_sums = xch + x_fields(has_gait ? 1 : 0) * kEnvBlock;   // [kSumRows][64]
    float* lds_aux = lds_sums + kSumRows * kEnvBlock;        // [kAuxRows][64]: 4 per float4 chunk of a DOF row
    constexpr int kPreRows = ws_pre_rows<P>(), kNormRows = ws_obs_norm_rows<P>();
    float* lds_pre = lds_aux + kAuxRows * kEnvBlock;         // [kPreRows][64]: values of the memory-only reward terms (wave 2 → wave 1)
    float* lds_norm = lds_pre + kPreRows * kEnvBlock;        // [kNormRows][64]: contact-force norms of observation items (wave 3, before the barrier)
    float* tile = lds_norm + kNormRows * kEnvBlock;          // [64][O+1]

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    GF_WSTAMP(1);
    const int lane = threadIdx.x & (GF_WAVE - 1);
    const int64_t N = UNI(a.num_envs);
    const int64_t tile_id = blockIdx.x;
    const int64_t n0 = tile_id * kEnvBlock;
    const int64_t n_raw = n0 + lane;
    const bool live = n_raw < N;
    const int64_t n = live ? n_raw : N - 1;
    const uint32_t e = (uint32_t)n;
    const uint32_t genv = e + UNI(a.env_offset);
    int D = 4 * DV;
    if constexpr (!P::kStatic) {
        if constexpr (P::kTail) D = UNI(a.num_dofs);
    }
    const uint32_t needs = UNI(a.needs);
    int n_term = 0, n_rew = 0, n_cmd = 0, n_obs = 0;
    if constexpr (P::kStatic) {
        n_term = P::n_term; n_rew = P::n_rew; n_cmd = P::n_cmd; n_obs = P::n_obs;
    } else {
        n_term = UNI(a.num_term); n_rew = UNI(a.num_rew); n_cmd = UNI(a.n_cmd); n_obs = UNI(a.n_obs);
    }
    const uint64_t seed = UNI(a.seed);
    // GF_POST_OBSERVE_ONLY (a run-time argument of every program): everything up to the reset has run as launches of their own — the
    // masks are inputs, no env is "done" as far as this launch is concerned, the observation waves do their part
    const bool obs_only = UNI(a.obs_only) != 0;
    constexpr int R = DV;
    const uint32_t ro = e * (uint32_t)D;
    GfStepStats* const k_stats = UNI(a.stats);
    GfStepStats* shard = k_stats ? stats_shard(k_stats) : nullptr;
    const bool logging = UNI(a.logging) != 0;
    float* const k_sums = UNI(a.episode_sums);
    float* const k_reward = UNI(a.reward);
    const bool has_reward = n_rew >= 0 && k_reward != nullptr;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#if GF_WS_AHEAD > 0
    {   // EXPERIMENT (tools/ab_build.sh -DGF_WS_AHEAD=K): the rows of the tile K workgroups ahead — same XCD when K % 8 == 0 —
        // are pulled towards the caches while this tile works: one LDS-DMA dword per lane and array (every line of that tile's block is
        // touched), no register, no wait.  (A loop over two tiles per workgroup, the other form of the pipeline, makes the compiler hoist the
        // descriptor's scalar loads out of the loop: SGPRs spill into VGPR lanes, 77 -> 137 VGPRs — not measurable as a pipeline.)
        const int64_t next_id = tile_id + GF_WS_AHEAD;
        if (next_id < (N + kEnvBlock - 1) / kEnvBlock) {
            const int64_t n2 = next_id * kEnvBlock + lane;
            const uint32_t e2 = (uint32_t)(n2 < N ? n2 : N - 1), ro2 = e2 * (uint32_t)D;
            float* const pf = lds + kArgVec * 4 + wave * kEnvBlock;
            auto warm = [&](const bool on, const float* base, const uint32_t off) GF_INLINE_LAMBDA {
                __builtin_amdgcn_global_load_lds(gsel(on, base, off), pf, 4, 0, 0);
            };
            if (wave == 0) {
                warm((needs & PN_QUAT) != 0, UNI(a.quat), 4u * e2);
                warm((needs & PN_POS) != 0, UNI(a.pos), 3u * e2);
                warm((needs & PN_LIN) != 0, UNI(a.lin_vel), 3u * e2);
                warm((needs & PN_ANG) != 0, UNI(a.ang_vel), 3u * e2);
                warm((needs & PN_EPLEN) != 0, reinterpret_cast<const float*>(UNI(a.episode_length)), e2);
                warm((needs & PN_MAXLEN) != 0, reinterpret_cast<const float*>(UNI(a.max_episode_length)), e2);
                if (has_gait) warm(true, UNI(a.gait.state), (uint32_t)GF_GAIT_ROW * e2);
            } else if (wave == 1) {
                warm((needs & PN_DOFDEV) != 0, UNI(a.dof_pos), ro2);
                warm((needs & PN_ACTRATE) != 0, UNI(a.env_actions), ro2);
                warm((needs & PN_ACTRATE) != 0, UNI(a.env_last_actions), ro2);
                warm(has_reward && UNI(a.episode_seconds) != nullptr, UNI(a.episode_seconds), e2);
            } else if (wave == 2) {
                warm((needs & PN_DOFPOS) != 0, UNI(a.dof_pos), ro2);
                warm((needs & PN_DOFVEL) != 0, UNI(a.dof_vel), ro2);
            } else {
                warm((needs & PN_TARGETS) != 0, UNI(a.targets), ro2);
                warm((needs & PN_ACTIONS) != 0, UNI(a.env_actions), ro2);
                warm(UNI(a.dof_force) != nullptr, UNI(a.dof_force), ro2);
            }
        }
    }
#endif

    // the command managers' table rows; a static program knows the widths
    auto cmd_row = [&](int c) GF_INLINE_LAMBDA {
        PostCmd cm = a.cmds[c];
        if constexpr (P::kStatic) cm.width = P::cmd_width[c];
        return cm;
    };

    // ---- contact phase: every ContactManager of the scene for this tile, in front of the phases that read their buffers -----------
    // (contact_manager.py:384-477, kernel.py:35-90 → gf_contact_tile.h; until round 4 a launch of its own.)  All four waves run it:
    // one lane per (env, tracked link), the tile's slot ids staged in LDS — the region the later phases use is free until then.  The
    // managers' public buffers are written to memory as before (they are API); the workgroup barrier behind the phase makes them
    // visible to this workgroup's waves (one CU, one L1), which read them back as L2 hits instead of as a dependent launch's loads.
    if constexpr (ws_prog_folds<P>()) {
        int fold_mgr = 0;
        if constexpr (P::kStatic) fold_mgr = karg.cfold.num_mgr;
        else fold_mgr = UNI(a.cfold.num_mgr);
        if (fold_mgr > 0) {
            int32_t* const cw = reinterpret_cast<int32_t*>(xch);
            int32_t* const t_mgr = cw;                                               // [kFoldMaxMgr][kContactMgrWords]
            int32_t* const t_target = t_mgr + kFoldMaxMgr * kContactMgrWords;        // [kFoldMaxTargets]
            uint16_t* const t_meta = reinterpret_cast<uint16_t*>(t_target + kFoldMaxTargets);   // [kFoldMaxTargets]
            int32_t* const t_with = t_target + kFoldMaxTargets + kFoldMaxTargets / 2; // [kFoldMaxMgr][GF_MAX_LINK_IDS]
            int32_t* const ids = cw + kFoldTableWords;
            // a word of the compact image at a lane-dependent offset (scalar loads at constant offsets plus per-lane selects were
            // measured slower: 5.5 vs 4.1 us to the end of the slot-id stage at 8 192 envs, profiles/r04_h_stamps.txt)
            auto cf_word = [&](int byte_off) GF_INLINE_LAMBDA -> uint32_t {
                if constexpr (P::kStatic)
                    return *(const __attribute__((address_space(4))) uint32_t*)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                                                               offsetof(GfPostArgs, cfold) + byte_off);
                else
                    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds) + offsetof(GfPostArgs, cfold) + byte_off);
            };
            const int tid = (int)threadIdx.x;
            const int T = UNI(a.cfold.total_targets), C = UNI(a.cfold.num_contacts);
            const int envs_here = (int)((N - n0) < kEnvBlock ? (N - n0) : kEnvBlock);
            const ContactScene sc{UNI(a.cfold.force), UNI(a.cfold.position), UNI(a.cfold.links_quat), UNI(a.cfold.links_vel), UNI(a.cfold.links_pos),
                                  UNI(a.cfold.link_a), UNI(a.cfold.link_b), C, UNI(a.cfold.num_scene_links), T, UNI(a.cfold.dt)};
            // the tile's slot ids are requested FIRST: they depend on nothing, and the table rows below are vector loads from the
            // kernel-argument segment — a cold round trip the slot ids now share instead of following
            const ContactIdsPending<4> pend = contact_ids_request<4>(sc, n0, envs_here, tid, kWsBlock);
            if (tid < fold_mgr) {   // this manager's image (ContactMgrL) and with-filter list
                const int base = (int)offsetof(PostContact, m) + tid * (int)sizeof(PostContactMgr);
                int32_t* img = t_mgr + tid * kContactMgrWords;
                const uint32_t packed = cf_word(base + 76);
                img[0] = (int32_t)(packed & 0xffu); img[1] = (int32_t)((packed >> 8) & 0xffu);
                img[2] = (int32_t)((packed >> 16) & 0xffu); img[3] = (int32_t)(packed >> 24);
                img[4] = (int32_t)cf_word(base + 72); img[5] = 0;
#pragma unroll
                for (int w = 0; w < 18; ++w) img[6 + w] = (int32_t)cf_word(base + 4 * w);
#pragma unroll
                for (int w = 0; w < kFoldMaxWith / 4; ++w) {
                    const uint32_t b4 = cf_word(base + 80 + 4 * w);
#pragma unroll
                    for (int j = 0; j < 4; ++j) t_with[tid * GF_MAX_LINK_IDS + 4 * w + j] = (int32_t)((b4 >> (8 * j)) & 0xffu);
                }
            }
            if (tid < T) {
                const int sh = 8 * (tid & 3), wd = tid & ~3;
                t_target[tid] = (int32_t)((cf_word((int)offsetof(PostContact, target_ids) + wd) >> sh) & 0xffu);
                t_meta[tid] = (uint16_t)(((cf_word((int)offsetof(PostContact, mgr_of) + wd) >> sh) & 0xffu) |
                                         (((cf_word((int)offsetof(PostContact, local_of) + wd) >> sh) & 0xffu) << 8));
            }
            if (N <= 16384) {   // (small launches only: with every workgroup of 65 536 envs resident the requested lines do not survive in the 4 MB L2 of an
                                // XCD until the roles ask for them — PMC reads of the gait task's launch 166 MB with the warm-up, profiles/r04_k_pmc_cfg.md)
                // While the contact phase runs its own chain of round trips, the rows the roles below will request are pulled towards this CU:
                // one LDS-DMA dword per lane and array (no register, no wait; the first dword of the env's row — rows are at most a
                // cache line long, so every line of the tile's block is touched) into a scratch row nobody reads.  The roles' real
                // loads then hit the L1 / L2 instead of starting a round trip to memory behind the phase.
                float* const pf = reinterpret_cast<float*>(ids + contact_lds_ints(kEnvBlock, C)) + wave * kEnvBlock;
                auto warm = [&](const bool on, const float* base, const uint32_t off) GF_INLINE_LAMBDA {
                    __builtin_amdgcn_global_load_lds(gsel(on, base, off), pf, 4, 0, 0);
                };
                if (wave == 0) {
                    warm((needs & PN_QUAT) != 0, UNI(a.quat), 4u * e);
                    warm((needs & PN_POS) != 0, UNI(a.pos), 3u * e);
                    warm((needs & PN_LIN) != 0, UNI(a.lin_vel), 3u * e);
                    warm((needs & PN_ANG) != 0, UNI(a.ang_vel), 3u * e);
                    warm((needs & PN_EPLEN) != 0, reinterpret_cast<const float*>(UNI(a.episode_length)), e);
                    warm((needs & PN_MAXLEN) != 0, reinterpret_cast<const float*>(UNI(a.max_episode_length)), e);
                    if (has_gait) warm(true, UNI(a.gait.state), (uint32_t)GF_GAIT_ROW * e);
                } else if (wave == 1) {
                    warm((needs & PN_DOFDEV) != 0, UNI(a.dof_pos), ro);
                    warm((needs & PN_ACTRATE) != 0, UNI(a.env_actions), ro);
                    warm((needs & PN_ACTRATE) != 0, UNI(a.env_last_actions), ro);
                    warm(has_reward && UNI(a.episode_seconds) != nullptr, UNI(a.episode_seconds), e);
                } else if (wave == 2) {
                    warm((needs & PN_DOFPOS) != 0, UNI(a.dof_pos), ro);
                    warm((needs & PN_DOFVEL) != 0, UNI(a.dof_vel), ro);
                } else {
                    warm((needs & PN_TARGETS) != 0, UNI(a.targets), ro);
                    warm((needs & PN_ACTIONS) != 0, UNI(a.env_actions), ro);
                    warm(UNI(a.dof_force) != nullptr, UNI(a.dof_force), ro);
                }
            }
            const ContactLds cl = contact_lds_carve(ids, kEnvBlock, C, reinterpret_cast<const ContactMgrL*>(t_mgr), t_target, t_meta, t_with);
            const int flag_mask = contact_tile<4, 4>(sc, cl, kEnvBlock, n0, envs_here, tid, kWsBlock, pend, [&](int i) GF_INLINE_LAMBDA { (void)i; GF_WSTAMP(12 + i); });   // (its first barrier covers the tables)
            if (shard) {   // non-finite force sanitised: the warning flag of contact_manager.py:399-403 (every folded manager counts into the step's block)
                const unsigned long long b = __ballot(flag_mask != 0);
                if (b && lane == 0) atomicOr(&shard->contact_flags, 1);
            }
            __syncthreads();   // the managers' buffers are visible to the tile's waves; the LDS is free for the phases below
        }
    }

    // ---- state that lives across the barrier, per role -------------------------------------------------------------
    // The roles are branches of ONE function, so to the register allocator a value that wave 0 carries across the barrier and a value
    // that wave 2 carries across it are both live at the barrier: role state ADDS UP.  The control wave's quaternion / counters / gait
    // row (25 words) and the three DOF rows the other waves hold (36 words at 12 DOF) therefore share storage — a union; every execution
    // path touches one member only and the compiler merges the two views word by word into the same registers.  The reward wave's rows
    // are dead once they are reduced, so what is left of them (position, two sums, seconds, three command columns) takes their place too,
    // written at the end of its pre-barrier block.  VGPRs: 12-DOF interpreter 148 → 116 (three → four workgroups per CU: 15.5 → 10.9 µs
    // at 65 536 envs, 171 → 145 µs at 1 M), 28-DOF interpreter 197 → 166 (two → three), gait program 104 → 86, the other static programs
    // 84–86 → 72.  What is indexed at run time by the interpreter (`cmd`, `cmd_dirty`) stays outside: inside the union such an access
    // becomes an address computation and the member goes to scratch.
    struct CtlState {                                   // wave 0
        __device__ CtlState() {}
        float4 q;
        int ep_len, term, trunc;
        float grow[GF_GAIT_ROW];                        // the gait manager's state row, stored after the last barrier
        int gait_sel, gait_resampled;
        uint32_t log_bits;                              // what the statistics count, taken behind the barrier: bit k = termination term k
        int gait_logged;                                //   fired, bit 16 + c = command c resampled; the gait selected before the reset
    };
    struct RowState { float4 r_a[R], r_b[R], r_c[R]; __device__ RowState() {} }; // wave 1: dof_pos/actions/last; wave 2: dof_pos/dof_vel/default; wave 3: targets/actions
    struct RewState {                                   // wave 1: what is left of its rows once they are reduced (written at the END of
        __device__ RewState() {}                        // its pre-barrier block, when it has read the rows for the last time)
        V3 pos;
        float dof_dev, act_rate, secs_in, cmd0[3];
        float vals[kPostMaxReward];                     // a static program's term values, computed BEFORE the barrier (see wave 1)
    };
    union Persist {
        CtlState ctl;
        RowState rows;
        RewState rew;
        __device__ Persist() {}
    } persist;
    RewState& rew = persist.rew;
    float4 (&r_a)[R] = persist.rows.r_a;
    float4 (&r_b)[R] = persist.rows.r_b;
    float4 (&r_c)[R] = persist.rows.r_c;
    CtlState& ctl = persist.ctl;
    float4& q = ctl.q;
    int& ep_len = ctl.ep_len; int& term = ctl.term; int& trunc = ctl.trunc;
    float cmd[GF_POST_MAX_CMD][kPostMaxRanges] = {};    // wave 0 (outside the union: indexed at run time by the interpreter)
    bool cmd_dirty[GF_POST_MAX_CMD] = {false, false};   // wave 0 (outside the union: a run-time index into a two-entry array)
    float (&grow)[GF_GAIT_ROW] = ctl.grow;
    int& gait_sel = ctl.gait_sel; int& gait_resampled = ctl.gait_resampled;
    if (wave == 0) {
        q = make_float4(1.f, 0.f, 0.f, 0.f);
        ep_len = 0; term = 0; trunc = 0;
#pragma unroll
        for (int j = 0; j < GF_GAIT_ROW; ++j) grow[j] = 0.f;
        gait_sel = 0; gait_resampled = 0;
        ctl.log_bits = 0u; ctl.gait_logged = 0;
    } else {
#pragma unroll
        for (int c = 0; c < DV; ++c) { r_a[c] = z4; r_b[c] = z4; r_c[c] = z4; }
    }

    // a static program's reward weights and sum rows, read from the kernel arguments BEFORE the barrier by the reward wave: uniform
    // values (scalar registers) — the fold behind the barrier then starts with them in hand instead of a scalar load per term
    float fold_w[kSumRows > 0 ? kSumRows : 1];
    int32_t fold_row[kSumRows > 0 ? kSumRows : 1];
    if (wave == 0) {
        // ---- control: loads -------------------------------------------------------------------------------------------
        q = ldg4(gsel((needs & PN_QUAT) != 0, UNI(a.quat), 4u * e));
        const GF_GLOBAL float* pp = gsel((needs & PN_POS) != 0, UNI(a.pos), 3u * e);
        const GF_GLOBAL float* lp = gsel((needs & PN_LIN) != 0, UNI(a.lin_vel), 3u * e);
        const GF_GLOBAL float* ap = gsel((needs & PN_ANG) != 0, UNI(a.ang_vel), 3u * e);
        const V3 pos{pp[0], pp[1], pp[2]};
        const V3 lin{lp[0], lp[1], lp[2]}, ang{ap[0], ap[1], ap[2]};
        ep_len = *gsel((needs & PN_EPLEN) != 0, UNI(a.episode_length), e);
        const int max_len = *gsel((needs & PN_MAXLEN) != 0, UNI(a.max_episode_length), e);
#pragma unroll
        for (int c = 0; c < GF_POST_MAX_CMD; ++c) {
            const bool on = c < n_cmd;
            const uint32_t w = on ? (uint32_t)UNI(cmd_row(c).width) : 0u;
            const GF_GLOBAL float* cp = gsel(on, on ? UNI(a.cmds[c].command) : nullptr, e * w);
#pragma unroll
            for (int j = 0; j < kPostMaxRanges; ++j) cmd[c][j] = cp[(uint32_t)j < w ? j : 0];
        }
        if (has_gait) {   // the gait state row: 4 x dwordx4, one wave = 4 KiB contiguous
            const GF_GLOBAL float* gp = G(UNI(a.gait.state)) + (int64_t)e * GF_GAIT_ROW;
            const float4 g0 = ldg4(gp), g1 = ldg4(gp + 4), g2 = ldg4(gp + 8), g3 = ldg4(gp + 12);
            grow[0] = g0.x; grow[1] = g0.y; grow[2] = g0.z; grow[3] = g0.w; grow[4] = g1.x; grow[5] = g1.y; grow[6] = g1.z; grow[7] = g1.w;
            grow[8] = g2.x; grow[9] = g2.y; grow[10] = g2.z; grow[11] = g2.w; grow[12] = g3.x; grow[13] = g3.y; grow[14] = g3.z; grow[15] = g3.w;
            gait_sel = (int)G(UNI(a.gait.selected))[e];
        }
        if (obs_only) {   // an env the reset of this tick touched is observed through its pre-reset quaternion (entity_manager.py:189-195)
            term = live ? (int)G(UNI(a.terminated))[n] : 0;
            trunc = live ? (int)G(UNI(a.truncated))[n] : 0;
            const float* const stash = UNI(a.quat_stash);
            if (stash && (needs & PN_QUAT) && (term | trunc)) q = ldg4(G(stash) + 4u * e);
        }
        // a static program's contact-count terminations: the counts are taken HERE, so the contact rows are requested together with
        // the loads above instead of as a round trip of their own between two terms' statistics branches
        int pre_cnt[ws_term_rows<P>()] = {};
        if constexpr (P::kStatic) {
            static_for<P::n_term>([&](auto K) GF_INLINE_LAMBDA {
                constexpr int k_ = decltype(K)::value;
                if constexpr (term_op_counts_contacts(P::term[k_].op))
                    pre_cnt[k_] = contact_count_over(a.contact[karg.tterms[k_].i[0]], n, karg.tterms[k_].p[0]);
            });
        }
        // ---- body-frame vectors, termination -----------------------------------------------------------------------------
        const V3 blin = rot_inv(q, lin), bang = rot_inv(q, ang), grav = rot_inv(q, V3{0.f, 0.f, -1.f});
        // the termination phase ran as a launch of its own (Python-level terms in the step): its masks are inputs.  A static program
        // says so in its signature (`term_done`, programs compiled at run time); the built-in ones all have their termination table
        if constexpr (!P::kStatic) {
            if (UNI(a.term_done)) {
                term = live ? (int)G(UNI(a.terminated))[n] : 0;
                trunc = live ? (int)G(UNI(a.truncated))[n] : 0;
            }
        } else if constexpr (prog_term_done<P>::value) {
            term = live ? (int)G(UNI(a.terminated))[n] : 0;
            trunc = live ? (int)G(UNI(a.truncated))[n] : 0;
        }
        TermRegs tr;
        const int has_maxlen = UNI(a.has_maxlen);
        tr.ep_len = ep_len; tr.max_len = max_len; tr.has_maxlen = has_maxlen != 0; tr.pos = pos; tr.m = n;
        tr.tilt_sin = clamp_max(norm2(grav.x, grav.y), 0.99f);
        auto term_body = [&](int k, const GfTerm& t, int v) GF_INLINE_LAMBDA {
            v = live ? v : 0;
            const int is_to = (t.flags & GF_TERM_FLAG_TIME_OUT) ? 1 : 0;   // (selects on values: an `if … trunc |= v; else term |= v;` makes the
            trunc |= is_to ? v : 0;                                        //  compiler pick an ADDRESS inside the union below → scratch)
            term |= is_to ? 0 : v;
            ctl.log_bits |= v ? (1u << k) : 0u;   // counted behind the barrier (this wave is the tile's longest before it, idle after)
        };
        if constexpr (P::kStatic) {
            static_for<P::n_term>([&](auto K) GF_INLINE_LAMBDA {
                constexpr int k_ = decltype(K)::value;
                GfTerm t = karg.tterms[k_];
                t.op = P::term[k_].op; t.flags = P::term[k_].flags;
                if constexpr (term_op_counts_contacts(P::term[k_].op)) term_body(k_, t, termination_from_count(t, pre_cnt[k_], ep_len));
                else term_body(k_, t, eval_termination_term(t, a, tr, (uint32_t)has_maxlen));
            });
        } else {
            for (int k = 0; k < n_term; ++k) {
                const GfTerm t = a.tterms[k];
                term_body(k, t, eval_termination_term(t, a, tr, (uint32_t)has_maxlen));
            }
        }
        const bool done0 = live && (term | trunc) && !obs_only && !UNI(a.no_reset);   // (GF_POST_NO_RESET: the masks are written, no env is treated as done)
        GF_WSTAMP(9);
        // ---- command.step then command.reset draws (values only; stores wait for the barrier) ------------------------------
#pragma unroll
        for (int c = 0; c < GF_POST_MAX_CMD; ++c) {
            if (c < n_cmd) {
                const PostCmd cm = cmd_row(c);
                const bool go = live && (ep_len % UNI(cm.resample_steps)) == 0;
                ctl.log_bits |= go ? (0x10000u << c) : 0u;
                if (go) {
                    const float4 u4 = draw_unit4(seed, cm.stream_step, genv, 0u);
                    const float nv[kPostMaxRanges] = {uniform_range(u4.x, cm.lo[0], cm.hi[0]), uniform_range(u4.y, cm.lo[1], cm.hi[1]),
                                                      uniform_range(u4.z, cm.lo[2], cm.hi[2]), uniform_range(u4.w, cm.lo[3], cm.hi[3])};
#pragma unroll
                    for (int j = 0; j < kPostMaxRanges; ++j)
                        if (j < cm.width) cmd[c][j] = nv[j];
                    cmd_dirty[c] = true;
                }
                if (done0) {
                    const float4 u4 = draw_unit4(seed, cm.stream_reset, genv, 0u);
                    const float nv[kPostMaxRanges] = {uniform_range(u4.x, cm.lo[0], cm.hi[0]), uniform_range(u4.y, cm.lo[1], cm.hi[1]),
                                                      uniform_range(u4.z, cm.lo[2], cm.hi[2]), uniform_range(u4.w, cm.lo[3], cm.hi[3])};
#pragma unroll
                    for (int j = 0; j < kPostMaxRanges; ++j)
                        if (j < cm.width) cmd[c][j] = nv[j];
                    cmd_dirty[c] = true;
                }
            }
        }
        GF_WSTAMP(10);
        // ---- GaitCommandManager.step, then .reset for done envs (gf_gait.hip: gait_body in both modes), on registers ------------
        if (has_gait) {
            const PostGait& gg = a.gait;
            const float two_pi = UNI(gg.two_pi), g_dt = UNI(gg.dt);
            const int num_gaits = UNI(gg.num_gaits), fixed_mask = UNI(gg.fixed_clearance_mask);
            auto resample = [&](const uint64_t stream) GF_INLINE_LAMBDA {   // resample_command -> _set_gait (:185-211, 347-377)
                const float4 u = draw_unit4(seed, stream, genv, 0u);
                int g = 0;
#pragma unroll
                for (int k = 0; k + 1 < GF_MAX_GAITS; ++k) g += (k + 1 < num_gaits && u.x >= gg.cum_weight[k]) ? 1 : 0;
                gait_sel = g;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    float o = gg.gait_offsets[0][f];
#pragma unroll
                    for (int k = 1; k < GF_MAX_GAITS; ++k) o = g == k ? gg.gait_offsets[k][f] : o;
                    grow[GF_GAIT_OFFSET + f] = o;
                }
                grow[GF_GAIT_HEIGHT] = ((fixed_mask >> g) & 1) ? gg.clearance_lo : uniform_range(u.y, gg.clearance_lo, gg.clearance_hi);
                grow[GF_GAIT_PERIOD] = uniform_range(u.z, gg.period_lo, gg.period_hi);
                gait_resampled = 1;
            };
            if (live && (ep_len % UNI(gg.resample_steps)) == 0) resample(gg.stream_step);
            ctl.gait_logged = gait_sel;   // _log_metrics (:430-441) counts envs per gait after this step's resample, before the reset's
            {   // the periodic clock (:231-239), for every env
                const float period = grow[GF_GAIT_PERIOD];
                const float gtime = torch_remainder(grow[GF_GAIT_TIME] + g_dt, period);
                const float phase = gtime / period;
                grow[GF_GAIT_TIME] = gtime;
                grow[GF_GAIT_PHASE] = phase;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const float fp = torch_remainder_one(phase + grow[GF_GAIT_OFFSET + f]);
                    sincos_det(two_pi * fp, &grow[GF_GAIT_CLOCK + f], &grow[GF_GAIT_CLOCK + 4 + f]);
                }
            }
            if (done0) {   // reset (:241-255): resample, zero the clock
                resample(gg.stream_reset);
#pragma unroll
                for (int j = 0; j < 8; ++j) grow[GF_GAIT_CLOCK + j] = 0.0f;
                grow[GF_GAIT_TIME] = 0.0f;
                grow[GF_GAIT_PHASE] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < GF_GAIT_ROW; ++j) xch[(X_GAIT + j) * kEnvBlock + lane] = grow[j];
        }
        GF_WSTAMP(11);
        // ---- publish ---------------------------------------------------------------------------------------------------------
        xch[X_TERM * kEnvBlock + lane] = (float)term;
        xch[X_TRUNC * kEnvBlock + lane] = (float)trunc;
        xch[X_DONE * kEnvBlock + lane] = done0 ? 1.f : 0.f;
        xch[(X_BLIN + 0) * kEnvBlock + lane] = blin.x; xch[(X_BLIN + 1) * kEnvBlock + lane] = blin.y; xch[(X_BLIN + 2) * kEnvBlock + lane] = blin.z;
        xch[(X_BANG + 0) * kEnvBlock + lane] = bang.x; xch[(X_BANG + 1) * kEnvBlock + lane] = bang.y; xch[(X_BANG + 2) * kEnvBlock + lane] = bang.z;
        xch[(X_GRAV + 0) * kEnvBlock + lane] = grav.x; xch[(X_GRAV + 1) * kEnvBlock + lane] = grav.y; xch[(X_GRAV + 2) * kEnvBlock + lane] = grav.z;
#pragma unroll
        for (int c = 0; c < GF_POST_MAX_CMD; ++c)
#pragma unroll
            for (int j = 0; j < kPostMaxRanges; ++j) xch[(X_CMD + c * kPostMaxRanges + j) * kEnvBlock + lane] = cmd[c][j];
        xch[X_DIRTY * kEnvBlock + lane] = (float)((cmd_dirty[0] ? 1 : 0) | (cmd_dirty[1] ? 2 : 0));
    } else if (wave == 1) {
        // ---- reward: loads + per-row reductions ------------------------------------------------------------------------------
        if (logging) {
            if constexpr (P::kStatic) {
                static_for<P::n_rew>([&](auto K) GF_INLINE_LAMBDA {
                    constexpr int k_ = decltype(K)::value;
                    __builtin_amdgcn_global_load_lds(k_sums + (int64_t)karg.rterms[k_].row * N + n, lds_sums + k_ * kEnvBlock, 4, 0, 0);
                });
            } else {
                for (int k = 0; k < n_rew; ++k)
                    __builtin_amdgcn_global_load_lds(k_sums + (int64_t)uni(a.rterms[k].row) * N + n, lds_sums + k * kEnvBlock, 4, 0, 0);
            }
        }
        const GF_GLOBAL float* p0 = gsel((needs & PN_DOFDEV) != 0, UNI(a.dof_pos), ro);
        const GF_GLOBAL float* p1 = gsel((needs & PN_ACTRATE) != 0, UNI(a.env_actions), ro);
        const GF_GLOBAL float* p2 = gsel((needs & PN_ACTRATE) != 0, UNI(a.env_last_actions), ro);
        const GF_GLOBAL float* p3 = gsel((needs & PN_DOFDEV) != 0, UNI(a.default_dof_pos), 0u);
        float4 r_def[R];
        row_load<DV>(r_a, p0, D); row_load<DV>(r_b, p1, D); row_load<DV>(r_c, p2, D); row_load<DV>(r_def, p3, D);
        const GF_GLOBAL float* pp = gsel((needs & PN_POS) != 0, UNI(a.pos), 3u * e);
        const V3 pos{pp[0], pp[1], pp[2]};
        // a static program whose terms read body-frame vectors: this wave derives them itself (see below) — the rows it needs for that
        // are requested here, with the tile's other loads
        float4 qb = make_float4(1.f, 0.f, 0.f, 0.f);
        V3 wlin{0.f, 0.f, 0.f}, wang{0.f, 0.f, 0.f};
        if constexpr (ws_rew_body_frame<P>()) {
            qb = ldg4(gsel((needs & PN_QUAT) != 0, UNI(a.quat), 4u * e));
            const GF_GLOBAL float* lp = gsel((needs & PN_LIN) != 0, UNI(a.lin_vel), 3u * e);
            const GF_GLOBAL float* ap = gsel((needs & PN_ANG) != 0, UNI(a.ang_vel), 3u * e);
            wlin = V3{lp[0], lp[1], lp[2]};
            wang = V3{ap[0], ap[1], ap[2]};
        }
        const float* k_secs = UNI(a.episode_seconds);
        const float secs_in = *gsel(has_reward && k_secs != nullptr, k_secs, e);
        float dof_dev = 0.f, act_rate = 0.f, cmd0[3];
        {   // command view 0 as it is BEFORE this step's resample (its new values are stored only after the second barrier)
            const float* v0 = UNI(a.command[0].command);
            const bool nv = v0 != nullptr;
            const uint32_t w = nv ? (uint32_t)UNI(a.command[0].width) : 0u;
            const uint32_t st = nv ? (uint32_t)(UNI(a.command[0].stride) ? UNI(a.command[0].stride) : UNI(a.command[0].width)) : 0u;
            const GF_GLOBAL float* cp = gsel(nv, v0, e * st);
            cmd0[0] = cp[0]; cmd0[1] = cp[w > 1 ? 1 : 0]; cmd0[2] = cp[w > 2 ? 2 : 0];
        }
#pragma unroll
        for (int c = 0; c < DV; ++c) {
            dof_dev += fabsf(r_a[c].x - r_def[c].x);
            dof_dev += fabsf(r_a[c].y - r_def[c].y);
            dof_dev += fabsf(r_a[c].z - r_def[c].z);
            dof_dev += fabsf(r_a[c].w - r_def[c].w);
        }
#pragma unroll
        for (int c = 0; c < DV; ++c) {
            float d;
            d = r_c[c].x - r_b[c].x; act_rate += d * d;
            d = r_c[c].y - r_b[c].y; act_rate += d * d;
            d = r_c[c].z - r_b[c].z; act_rate += d * d;
            d = r_c[c].w - r_b[c].w; act_rate += d * d;
        }
        // A static program evaluates its terms HERE, before the barrier: a term needs this wave's own reductions, memory nothing in
        // the launch writes before its last barrier, and at most the body-frame vectors — which this wave derives itself from the same
        // quaternion / velocity rows with the same rot_inv the control wave uses (same inputs, same arithmetic: the same bits) instead of
        // waiting for them at the barrier.  Behind the barrier only the fold is left (sums, statistics, stores): the wave used to sit
        // idle here until the control wave was done and then evaluate every term with the other three waiting for it at the tile
        // barrier (profiles/r03_n_stamps.txt: 2.7 of 10.5 µs at 8 192 envs).  The two terms that read the termination mask stay behind.
        float vals[kPostMaxReward];
        GF_WSTAMP(9);
        if constexpr (P::kStatic) {
            if constexpr (P::n_rew > 0) {   // (a program with reward terms matches only a launch that has the reward manager: no branch)
                RewardRegs rp;
                rp.pos = pos; rp.blin = V3{0.f, 0.f, 0.f}; rp.bang = rp.blin; rp.grav = rp.blin;
                if constexpr (ws_rew_body_frame<P>()) {
                    rp.blin = rot_inv(qb, wlin);
                    rp.bang = rot_inv(qb, wang);
                    rp.grav = rot_inv(qb, V3{0.f, 0.f, -1.f});
                }
                rp.dof_dev = dof_dev; rp.act_rate = act_rate; rp.terminated = 0;
                rp.cmd0[0] = cmd0[0]; rp.cmd0[1] = cmd0[1]; rp.cmd0[2] = cmd0[2];
                rp.n = n; rp.live = live;
                auto term_of = [&](auto K) GF_INLINE_LAMBDA {
                    constexpr int k_ = decltype(K)::value;
                    GfTerm t = karg.rterms[k_];
                    t.op = P::rew[k_].op; t.flags = P::rew[k_].flags; t.i[0] = P::rew[k_].i0; t.i[1] = P::rew[k_].i1;
                    return t;
                };
                static_for<P::n_rew>([&](auto K) GF_INLINE_LAMBDA {   // the stateful term (body acceleration stores its state) goes last
                    constexpr int k_ = decltype(K)::value;
                    constexpr int op = P::rew[k_].op;
                    if constexpr (!reward_op_memory_only(op) && !reward_op_reads_terminated(op) && op != GF_R_BODY_ACCEL_EXP)
                        vals[k_] = eval_reward_term(term_of(K), a, rp);
                });
                static_for<P::n_rew>([&](auto K) GF_INLINE_LAMBDA {
                    constexpr int k_ = decltype(K)::value;
                    if constexpr (P::rew[k_].op == GF_R_BODY_ACCEL_EXP) vals[k_] = eval_reward_term(term_of(K), a, rp);
                });
            }
        }
        // the rows have been read for the last time: what is left of them takes their place in the union
        rew.pos = pos;
        rew.dof_dev = dof_dev; rew.act_rate = act_rate; rew.secs_in = secs_in;
        rew.cmd0[0] = cmd0[0]; rew.cmd0[1] = cmd0[1]; rew.cmd0[2] = cmd0[2];
        if constexpr (P::kStatic) {
            static_for<P::n_rew>([&](auto K) GF_INLINE_LAMBDA {
                constexpr int k_ = decltype(K)::value;
                rew.vals[k_] = vals[k_];
                fold_w[k_] = karg.rterms[k_].w; fold_row[k_] = karg.rterms[k_].row;
                asm volatile("" : "+s"(fold_w[k_]), "+s"(fold_row[k_]));   // (loaded HERE: not sunk to their use behind the barrier)
            });
        }
    } else if (wave == 2) {
        const GF_GLOBAL float* p0 = gsel((needs & PN_DOFPOS) != 0, UNI(a.dof_pos), ro);
        const GF_GLOBAL float* p1 = gsel((needs & PN_DOFVEL) != 0, UNI(a.dof_vel), ro);
        const float* k_def = UNI(a.default_dof_pos);
        const GF_GLOBAL float* p2 = gsel(k_def != nullptr, k_def, 0u);
        row_load<DV>(r_a, p0, D); row_load<DV>(r_b, p1, D); row_load<DV>(r_c, p2, D);
        GF_WSTAMP(9);
        if constexpr (kPreRows > 0) {   // the memory-only reward terms (see reward_op_memory_only), parked for the fold
            RewardRegs rp;   // (parked rows exist only in a program with reward terms, which matches only a launch that has the reward manager)
            rp.pos = V3{0.f, 0.f, 0.f}; rp.blin = rp.pos; rp.bang = rp.pos; rp.grav = rp.pos;
            rp.dof_dev = 0.f; rp.act_rate = 0.f; rp.terminated = 0;
            rp.cmd0[0] = 0.f; rp.cmd0[1] = 0.f; rp.cmd0[2] = 0.f;
            rp.n = n; rp.live = live;
            static_for<P::n_rew>([&](auto K) GF_INLINE_LAMBDA {
                constexpr int k_ = decltype(K)::value;
                if constexpr (reward_op_memory_only(P::rew[k_].op)) {
                    GfTerm t = karg.rterms[k_];
                    t.op = P::rew[k_].op; t.flags = P::rew[k_].flags; t.i[0] = P::rew[k_].i0; t.i[1] = P::rew[k_].i1;
                    lds_pre[ws_pre_slot<P>(k_) * kEnvBlock + lane] = eval_reward_term(t, a, rp);
                }
            });
        }
    } else {
        const GF_GLOBAL float* p0 = gsel((needs & PN_TARGETS) != 0, UNI(a.targets), ro);
        const GF_GLOBAL float* p1 = gsel((needs & PN_ACTIONS) != 0, UNI(a.env_actions), ro);
        row_load<DV>(r_a, p0, D); row_load<DV>(r_b, p1, D);
        if constexpr (P::kStatic && ws_obs_has<P>(GF_O_DOF_FORCE)) {   // the dof_force row: this wave's third row set is free
            row_load<DV>(r_c, gsel(UNI(a.dof_force) != nullptr, UNI(a.dof_force), ro), D);
        }
        if constexpr (kNormRows > 0) {   // contact-force norms of observation items
            static_for<P::n_obs>([&](auto M) GF_INLINE_LAMBDA {
                constexpr int m_ = decltype(M)::value;
                static_for<P::obs_items[m_]>([&](auto I) GF_INLINE_LAMBDA {
                    constexpr int i_ = decltype(I)::value;
                    if constexpr (P::item[m_][i_].op == GF_O_CONTACT_FORCE_NORM) {
                        const GfContactView cv = a.contact[P::item[m_][i_].i0];
                        const GF_GLOBAL float* r = G(cv.contacts) + n * cv.num_links * 3;
#pragma unroll
                        for (int l = 0; l < P::item[m_][i_].width; ++l)
                            lds_norm[(ws_obs_norm_slot<P>(m_, i_) + l) * kEnvBlock + lane] = norm3(r[3 * l], r[3 * l + 1], r[3 * l + 2]);
                    }
                });
            });
        }
    }
    GF_WSTAMP(2);
    // every load above has landed (registers / LDS) before any wave starts storing state behind the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    GF_WSTAMP(3);

    const bool done = xch[X_DONE * kEnvBlock + lane] != 0.f;
    const unsigned long long done_mask = __ballot(done);
    const int scene_reset = UNI(a.scene_reset), zero_velocity = UNI(a.zero_velocity), reset_env = UNI(a.reset_env);

    if (wave == 0) {
        // ---- control: stores --------------------------------------------------------------------------------------------------
        if (live && !obs_only) {
            G(UNI(a.terminated))[n_raw] = (uint8_t)term;
            G(UNI(a.truncated))[n_raw] = (uint8_t)trunc;
            uint8_t* const rd = UNI(a.roll_done);
            if (rd) G(rd)[n_raw] = (uint8_t)((term | trunc) != 0);   // dones[t] of the rollout storage
        }
        if (shard && done_mask && lane == 0) atomicAdd(&shard->reset_count, popc64(done_mask));
        if (shard) {   // the step's counters, from the bits the pre-barrier block left
            auto count_bits = [&](int k, int32_t* dst) GF_INLINE_LAMBDA {
                const unsigned long long hit = __ballot((ctl.log_bits >> k) & 1u);
                if (hit && lane == 0) atomicAdd(dst, popc64(hit));
            };
            if constexpr (P::kStatic) {
                static_for<P::n_term>([&](auto K) GF_INLINE_LAMBDA { count_bits(decltype(K)::value, &shard->term_fired[decltype(K)::value]); });
            } else {
                for (int k = 0; k < n_term; ++k) count_bits(k, &shard->term_fired[k]);
            }
#pragma unroll
            for (int c = 0; c < GF_POST_MAX_CMD; ++c)
                if (c < n_cmd) count_bits(16 + c, &shard->resample_count);
            if (has_gait) {
#pragma unroll
                for (int g = 0; g < GF_MAX_GAITS; ++g) {
                    const unsigned long long b = __ballot(live && ctl.gait_logged == g);
                    if (b && lane == 0) atomicAdd(&shard->gait_count[g], popc64(b));
                }
            }
        }
        if (has_gait) {
            uint8_t* const fout = UNI(a.gait.flags_out);
            if (fout) {   // this block's "any env in swing / stance" byte for the state the launch leaves — into the OTHER buffer
                const float two_pi = UNI(a.gait.two_pi), pi = 0.5f * two_pi;
                uint32_t byte = 0;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int fl = gait_foot_flags(grow[GF_GAIT_PHASE], grow[GF_GAIT_OFFSET + f], two_pi, pi);
                    if (__ballot(live && (fl & 1))) byte |= 1u << (2 * f);
                    if (__ballot(live && (fl & 2))) byte |= 2u << (2 * f);
                }
                if (lane == 0) G(fout)[tile_id] = (uint8_t)byte;
            }
        }
        if (done) {
            if ((reset_env & 1) && a.env_actions) {
                float4 zr[R];
#pragma unroll
                for (int c = 0; c < DV; ++c) zr[c] = z4;
                row_store<DV>(G(a.env_actions) + n * D, zr, D);
                row_store<DV>(G(a.env_last_actions) + n * D, zr, D);
            }
            if (reset_env & 2) G(a.episode_length)[n] = 0;
            if (a.max_episode_length && a.max_random_scaling > 0.0f) {
                const float u = draw_unit4(seed, a.stream_reset, genv, 0u).x;
                const float rnd = uniform_range(u, -1.0f, 1.0f) * a.max_random_scaling;
                G(a.max_episode_length)[n] = (int32_t)rintf((float)a.base_max_episode_length + rnd);
            }
            if (scene_reset) {
                GF_GLOBAL float* wp = G(a.pos) + 3 * n;
                float np[3] = {a.reset_pos[0], a.reset_pos[1], a.reset_pos[2]};
                float4 nq = make_float4(a.reset_quat[0], a.reset_quat[1], a.reset_quat[2], a.reset_quat[3]);
                bool set_quat = a.set_quat != 0;
                if (UNI(a.spawn_mode)) {  // mdp.reset.randomize_terrain_position
                    float u[5];
                    spawn_draws(nullptr, n, seed, a.stream_reset, genv, a.spawn_rot_mask, u);
                    spawn_pose(a, u, np, &nq);
                    set_quat = a.spawn_set_quat != 0;
                }
                wp[0] = np[0]; wp[1] = np[1]; wp[2] = np[2];
                if (set_quat) {
                    if (a.quat_stash) reinterpret_cast<GF_GLOBAL f32x4*>(G(a.quat_stash))[n] = f32x4{q.x, q.y, q.z, q.w};
                    reinterpret_cast<GF_GLOBAL f32x4*>(G(a.quat))[n] = f32x4{nq.x, nq.y, nq.z, nq.w};
                }
                if (zero_velocity) {
                    GF_GLOBAL float* wl = G(a.lin_vel) + 3 * n;
                    GF_GLOBAL float* wa = G(a.ang_vel) + 3 * n;
                    wl[0] = 0.f; wl[1] = 0.f; wl[2] = 0.f;
                    wa[0] = 0.f; wa[1] = 0.f; wa[2] = 0.f;
                }
            }
        }
    } else if (wave == 1) {
        // ---- reward: the term fold (reward_manager.py:166-195) + the manager's reset folded into the sum update -----------------
        if (has_reward) {
            RewardRegs rr;
            rr.pos = rew.pos;
            rr.blin = V3{xch[(X_BLIN + 0) * kEnvBlock + lane], xch[(X_BLIN + 1) * kEnvBlock + lane], xch[(X_BLIN + 2) * kEnvBlock + lane]};
            rr.bang = V3{xch[(X_BANG + 0) * kEnvBlock + lane], xch[(X_BANG + 1) * kEnvBlock + lane], xch[(X_BANG + 2) * kEnvBlock + lane]};
            rr.grav = V3{xch[(X_GRAV + 0) * kEnvBlock + lane], xch[(X_GRAV + 1) * kEnvBlock + lane], xch[(X_GRAV + 2) * kEnvBlock + lane]};
            rr.dof_dev = rew.dof_dev; rr.act_rate = rew.act_rate; rr.terminated = xch[X_TERM * kEnvBlock + lane] != 0.f ? 1 : 0;
            rr.cmd0[0] = rew.cmd0[0]; rr.cmd0[1] = rew.cmd0[1]; rr.cmd0[2] = rew.cmd0[2];
            rr.n = n; rr.live = live;
            const float dt = UNI(a.dt);
            const uint32_t log_mask = UNI(a.reward_log_mask);
            const float secs_new = rew.secs_in + dt;
            float buf = 0.f;
            GF_WSTAMP(10);
            const bool log_reset = logging && done_mask != 0;
            auto fold_body = [&](int k, const float t_w, const int32_t t_row, float v) GF_INLINE_LAMBDA {
                v = v * t_w;
                buf += v;
                if (logging) {
                    float s = lds_sums[k * kEnvBlock + lane] + v;
                    if (log_reset) {
                        const float per_sec = done ? s / secs_new : 0.f;
                        if (shard && (log_mask & (1u << t_row))) {
                            if (popc64(done_mask) > 4) {
                                const double w = wave_sum((double)per_sec);
                                if (lane == 0) unsafeAtomicAdd(&shard->reward_episode_sum[t_row], w);
                            } else if (done) {
                                unsafeAtomicAdd(&shard->reward_episode_sum[t_row], (double)per_sec);
                            }
                        }
                        if (done) s = 0.f;
                    }
                    if (live) G(k_sums)[(int64_t)t_row * N + n_raw] = s;
                }
            };
            if constexpr (P::kStatic) {
                // The values were computed before the barrier — by this wave (everything it can evaluate from its own loads) and by
                // wave 2 (the memory-only terms, parked in LDS rows); the two terms that read the termination mask are evaluated here.
                // What is left is the fold: the reference's order, the same arithmetic per term, with the sum / statistics stores.
                auto term_of = [&](auto K) GF_INLINE_LAMBDA {
                    constexpr int k_ = decltype(K)::value;
                    GfTerm t = karg.rterms[k_];
                    t.op = P::rew[k_].op; t.flags = P::rew[k_].flags; t.i[0] = P::rew[k_].i0; t.i[1] = P::rew[k_].i1;
                    return t;
                };
                static_for<P::n_rew>([&](auto K) GF_INLINE_LAMBDA {
                    constexpr int k_ = decltype(K)::value;
                    constexpr int op = P::rew[k_].op;
                    float v;
                    if constexpr (reward_op_memory_only(op)) v = lds_pre[ws_pre_slot<P>(k_) * kEnvBlock + lane];
                    else if constexpr (reward_op_reads_terminated(op)) v = eval_reward_term(term_of(K), a, rr);
                    else v = rew.vals[k_];
                    fold_body(k_, fold_w[k_], fold_row[k_], v);
                });
            } else {
                for (int k = 0; k < n_rew; ++k) {
                    const GfTerm t = a.rterms[k];
                    fold_body(k, t.w, t.row, eval_reward_term(t, a, rr));
                }
            }
            GF_WSTAMP(11);
            if (live) {
                G(k_reward)[n_raw] = buf;
                G(UNI(a.episode_seconds))[n_raw] = done ? 1e-10f : secs_new;
                float* const rr_out = UNI(a.roll_reward);
                if (rr_out) G(rr_out)[n_raw] = buf;   // rewards[t] of the rollout storage
            }
            if (done && logging)
                for (int row = 0; row < a.reward_rows; ++row)
                    if (a.uncovered_rows & (1u << row)) G(k_sums)[(int64_t)row * N + n_raw] = 0.f;
        }
    } else if (wave == 2) {
        // ---- DOF reset of done envs: memory and the registers the observation reads (position_action_manager.py:455-464) --------
        if (done) {
            float* const k_dvel = UNI(a.dof_vel);
            if (UNI(a.reset_dofs)) {
                const float dof_noise = a.dof_noise_scale;
                if (dof_noise != 0.0f) {
#pragma nounroll
                    for (int c = 0; c < DV; ++c) {
                        const float4 r4 = draw_unit4(seed, a.stream_reset, genv, (uint32_t)(1 + c));
                        float* sc = lds_aux + (4 * c) * kEnvBlock + lane;
                        sc[0 * kEnvBlock] = uniform_range(r4.x, -1.0f, 1.0f) * dof_noise;
                        sc[1 * kEnvBlock] = uniform_range(r4.y, -1.0f, 1.0f) * dof_noise;
                        sc[2 * kEnvBlock] = uniform_range(r4.z, -1.0f, 1.0f) * dof_noise;
                        sc[3 * kEnvBlock] = uniform_range(r4.w, -1.0f, 1.0f) * dof_noise;
                    }
                }
#pragma unroll
                for (int c = 0; c < DV; ++c) {
                    float4 p = r_c[c];
                    if (dof_noise != 0.0f) {
                        const float* sc = lds_aux + (4 * c) * kEnvBlock + lane;
                        p.x = p.x + sc[0 * kEnvBlock];
                        p.y = p.y + sc[1 * kEnvBlock];
                        p.z = p.z + sc[2 * kEnvBlock];
                        p.w = p.w + sc[3 * kEnvBlock];
                    }
                    r_a[c] = p;
                    if (k_dvel) r_b[c] = z4;
                }
                row_store<DV>(G(a.dof_pos) + n * D, r_a, D);   // (a row's last chunk: only its own floats, row_store)
                if (k_dvel) row_store<DV>(G(k_dvel) + n * D, r_b, D);
            }
            if (scene_reset && zero_velocity && k_dvel) {
#pragma unroll
                for (int c = 0; c < DV; ++c) r_b[c] = z4;
                row_store<DV>(G(k_dvel) + n * D, r_b, D);
            }
        }
    } else {
        if (done && (reset_env & 1) && a.env_actions) {
#pragma unroll
            for (int c = 0; c < DV; ++c) r_b[c] = z4;  // raw actions of a reset env read as zero
        }
    }
    GF_WSTAMP(4);

    // ---- history shift (H > 1) of every observation manager: it depends on nothing this launch computes, and the reward wave's
    //      fold is the longest post-barrier role (a chain of per-term loads) — so waves 0, 2 and 3 move the pure history units
    //      (gf_obs_hist.h) while wave 1 folds; the frame and the units that touch it follow from the LDS tile below
    auto hist_early = [&](int m, const PostObs& ob, int O, int H) GF_INLINE_LAMBDA {
        float* const ob_out = UNI(ob.obs);
        const bool flat = H > 1 && !UNI(ob.ring) && O >= 4 && (reinterpret_cast<uintptr_t>(ob_out) & 15u) == 0;
        if (!flat || wave == 1) return;
        const int rows = (int)((N - n0) < kEnvBlock ? (N - n0) : kEnvBlock);
        const int64_t OH = (int64_t)O * H;
        GF_GLOBAL float* out = G(ob_out) + n0 * OH;
        const GF_GLOBAL float* prev = G(UNI(ob.prev)) + n0 * OH;
        float* const roll_base = UNI(a.roll_obs);
        GF_GLOBAL float* roll = (roll_base && UNI(a.roll_obs_index) == m) ? G(roll_base) + n0 * OH : nullptr;
        const int units = (rows * (int)OH) >> 2;
        const FastDiv dr((int)OH);
        constexpr int kLanes = 3 * kEnvBlock;
        const int lid = (wave == 0 ? 0 : wave - 1) * kEnvBlock + lane;
        HistBatch hb;
        if ((UNI(a.obs_stream) >> m) & 1u) {   // kObsStreamBytes (gf_post_args.h)
            for (int first = lid; first - lid < units; first += kObsShift * kLanes) {   // wave-uniform trip count
                hist_load<true>(hb, prev, first, units, O, (int)OH, dr, kLanes);
                hist_store<true>(hb, out, first, kLanes);
                if (roll) hist_store<true>(hb, roll, first, kLanes, false);
            }
        } else {
            for (int first = lid; first - lid < units; first += kObsShift * kLanes) {
                hist_load(hb, prev, first, units, O, (int)OH, dr, kLanes);
                hist_store(hb, out, first, kLanes);
                if (roll) hist_store(hb, roll, first, kLanes, false);
            }
        }
    };
    if constexpr (P::kStatic) {
        static_for<P::n_obs>([&](auto M) GF_INLINE_LAMBDA {
            constexpr int m_ = decltype(M)::value;
            if constexpr (P::obs_history[m_] > 1) hist_early(m_, karg.obs[m_], P::obs_width[m_], P::obs_history[m_]);
        });
    } else {
        for (int m = 0; m < n_obs; ++m) hist_early(m, a.obs[m], uni(a.obs[m].width), uni(a.obs[m].history));
    }

    // ---- observations: waves 2 and 3 assemble the tile, all four stream it out ---------------------------------------------------
    auto obs_manager = [&](int m, const PostObs& ob, int O, int H, int n_items, auto&& each_item) GF_INLINE_LAMBDA {
        const int S = O + 1;
        float* const ob_out = UNI(ob.obs);
        const float* const ob_prev = UNI(ob.prev);
        const uint64_t ob_stream = UNI(ob.stream);
        if (wave >= 2) {
            float* row = tile + lane * S;
            const bool rows_wave = wave == 2;
            const bool zeroed = done && scene_reset && zero_velocity;
            int col = 0;
            auto item_body = [&](const GfObsItem& it, const bool scaled, const bool noisy, const int norm_slot) GF_INLINE_LAMBDA {
                const int op = it.op, it_w = it.width;
                const ObsFin f{it.scale, scaled};
                const bool row_item = op == GF_O_DOF_POS || op == GF_O_DOF_VEL;
                if (row_item == rows_wave) {
                    switch (op) {
                        case GF_O_DOF_POS: put_row<DV>(f, r_a, row, col, D); break;
                        case GF_O_DOF_VEL: put_row<DV>(f, r_b, row, col, D); break;
                        case GF_O_ACTIONS: put_row<DV>(f, r_a, row, col, D); break;
                        case GF_O_RAW_ACTIONS: put_row<DV>(f, r_b, row, col, D); break;
                        case GF_O_COMMAND: {
                            const int owner = a.cmd_of_view[it.i0];
                            if (owner == kViewGait) {   // the gait manager's post-step, post-reset row (observation(): columns 0..13)
#pragma unroll
                                for (int j = 0; j < GF_GAIT_ROW; ++j)
                                    if (j < it_w) row[col + j] = obs_finish(f, xch[(X_GAIT + j) * kEnvBlock + lane], col + j);
                            } else if (owner >= 0) {
#pragma unroll
                                for (int j = 0; j < kPostMaxRanges; ++j)
                                    if (j < it_w) row[col + j] = obs_finish(f, xch[(X_CMD + owner * kPostMaxRanges + j) * kEnvBlock + lane], col + j);
                            } else {
                                const GfCommandView cv = a.command[it.i0];
                                for (int j = 0; j < it_w; ++j) row[col + j] = obs_finish(f, G(cv.command)[n * cmd_stride(cv) + j], col + j);
                            }
                        } break;
                        case GF_O_ANG_VEL_BODY:
                        case GF_O_LIN_VEL_BODY:
                        case GF_O_PROJ_GRAVITY: {
                            const int base = op == GF_O_ANG_VEL_BODY ? X_BANG : (op == GF_O_LIN_VEL_BODY ? X_BLIN : X_GRAV);
                            const bool z = zeroed && op != GF_O_PROJ_GRAVITY;  // rot_inv(q, 0) is exactly +0
                            row[col + 0] = obs_finish(f, z ? 0.f : xch[(base + 0) * kEnvBlock + lane], col + 0);
                            row[col + 1] = obs_finish(f, z ? 0.f : xch[(base + 1) * kEnvBlock + lane], col + 1);
                            row[col + 2] = obs_finish(f, z ? 0.f : xch[(base + 2) * kEnvBlock + lane], col + 2);
                        } break;
                        case GF_O_DOF_FORCE: {
                            if constexpr (P::kStatic) {   // loaded with the tile's other rows, before the barrier (wave 3's third row set)
                                put_row<DV>(f, r_c, row, col, D);
                            } else {
                                const GF_GLOBAL float* r = G(a.dof_force) + n * D;
                                for (int j = 0; j < it_w; ++j) row[col + j] = obs_finish(f, r[j], col + j);
                            }
                        } break;
                        case GF_O_CONTACT_FORCE_NORM: {
                            if constexpr (P::kStatic) {   // parked by this wave before the barrier
                                for (int l = 0; l < it_w; ++l) row[col + l] = obs_finish(f, lds_norm[(norm_slot + l) * kEnvBlock + lane], col + l);
                            } else {
                                const GfContactView cv = a.contact[it.i0];
                                const GF_GLOBAL float* r = G(cv.contacts) + n * cv.num_links * 3;
                                for (int l = 0; l < it_w; ++l) row[col + l] = obs_finish(f, norm3(r[3 * l], r[3 * l + 1], r[3 * l + 2]), col + l);
                            }
                        } break;
                        default: break;
                    }
                    if (noisy) {   // column c draws word c & 3 of Philox block c >> 2: one block per four columns the item covers
                        const float it_noise = it.noise;
                        const int c_end = col + it_w;
#pragma nounroll
                        for (int b = col >> 2; b <= ((c_end - 1) >> 2); ++b) {
                            const float4 u4 = draw_unit4(seed, ob_stream, genv, (uint32_t)b);
                            const float uu[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
                            for (int w4 = 0; w4 < 4; ++w4) {
                                const int c = 4 * b + w4;
                                if (c >= col && c < c_end) row[c] = row[c] + uniform_range(uu[w4], -1.0f, 1.0f) * it_noise;
                            }
                        }
                    }
                }
                col += it_w;
            };
            each_item(item_body);
        }
        GF_WSTAMP(6);
        __syncthreads();
        GF_WSTAMP(7);
        {
            const int rows = (int)((N - n0) < kEnvBlock ? (N - n0) : kEnvBlock);
            const int64_t OH = (int64_t)O * H;
            const int ring = UNI(ob.ring);   // in-place history ring: only the new frame is written, into its slot
            const int ring_slots = UNI(ob.ring_slots);
            const int64_t OS = (ring && ring_slots) ? (int64_t)O * ring_slots : OH;   // env stride of `out` (a ring with more slots than frames)
            GF_GLOBAL float* out = G(ob_out) + n0 * OS + (ring ? (int64_t)(ring - 1) * O : 0);
            float* const roll_base = UNI(a.roll_obs);
            GF_GLOBAL float* roll = (roll_base && UNI(a.roll_obs_index) == m) ? G(roll_base) + n0 * OH : nullptr;   // observations[t+1] of the rollout storage
            const int t = threadIdx.x;
            // history (H > 1): 16-byte units over the tile's contiguous [rows, O·H] run, whatever O is (gf_obs_hist.h) — the
            // gait task's 62-wide policy frame has no 16-byte aligned rows, the run has
            const bool flat = H > 1 && !ring && O >= 4 && (reinterpret_cast<uintptr_t>(ob_out) & 15u) == 0;
            const bool nt = ((UNI(a.obs_stream) >> m) & 1u) != 0;   // kObsStreamBytes (gf_post_args.h)
            if (flat) {   // the pure history units went out above (hist_early); what touches the new frame comes from the tile
                const GF_GLOBAL float* prev = G(ob_prev) + n0 * OH;
                if (nt) write_mixed_units<true>(out, prev, tile, S, rows, O, (int)OH, t, roll);
                else write_mixed_units(out, prev, tile, S, rows, O, (int)OH, t, roll);
            } else if ((O & 3) == 0) {
                const int o4 = O >> 2;
                const int qstep = kWsBlock / o4, rstep = kWsBlock - qstep * o4;
                int rw = t / o4, c4 = t - rw * o4;
                for (int i = t; i < rows * o4; i += kWsBlock) {
                    const float* r = tile + rw * S + c4 * 4;
                    const f32x4 v4{r[0], r[1], r[2], r[3]};
                    if (nt) {
                        __builtin_nontemporal_store(v4, reinterpret_cast<GF_GLOBAL f32x4*>(out + rw * OS) + c4);
                        if (roll) __builtin_nontemporal_store(v4, reinterpret_cast<GF_GLOBAL f32x4*>(roll + rw * OH) + c4);
                    } else {
                        reinterpret_cast<GF_GLOBAL f32x4*>(out + rw * OS)[c4] = v4;
                        if (roll) reinterpret_cast<GF_GLOBAL f32x4*>(roll + rw * OH)[c4] = v4;
                    }
                    rw += qstep; c4 += rstep;
                    if (c4 >= o4) { c4 -= o4; ++rw; }
                }
                if (H > 1 && !ring) {
                    const int h4 = (O * (H - 1)) >> 2;
                    const GF_GLOBAL float* prev = G(ob_prev) + n0 * OH;
                    for (int i = t; i < rows * h4; i += kWsBlock) {
                        const int rw2 = i / h4, j = i - rw2 * h4;
                        const f32x4 hv = reinterpret_cast<const GF_GLOBAL f32x4*>(prev + rw2 * OH)[j];
                        reinterpret_cast<GF_GLOBAL f32x4*>(out + rw2 * OH + O)[j] = hv;
                        if (roll) reinterpret_cast<GF_GLOBAL f32x4*>(roll + rw2 * OH + O)[j] = hv;
                    }
                }
            } else {
                for (int i = t; i < rows * O; i += kWsBlock) {
                    const int rw = i / O, cc = i - rw * O;
                    out[rw * OS + cc] = tile[rw * S + cc];
                    if (roll) roll[rw * OH + cc] = tile[rw * S + cc];
                }
                if (H > 1 && !ring) {
                    const int hw = O * (H - 1);
                    const GF_GLOBAL float* prev = G(ob_prev) + n0 * OH;
                    for (int i = t; i < rows * hw; i += kWsBlock) {
                        const int rw = i / hw, j = i - rw * hw;
                        const float hv = prev[rw * OH + j];
                        out[rw * OH + O + j] = hv;
                        if (roll) roll[rw * OH + O + j] = hv;
                    }
                }
            }
        }
        if (m + 1 < n_obs) __syncthreads();  // the tile is reused by the next observation manager
    };
    GF_WSTAMP(5);
    if constexpr (P::kStatic) {
        static_for<P::n_obs>([&](auto M) GF_INLINE_LAMBDA {
            constexpr int m_ = decltype(M)::value;
            obs_manager(m_, karg.obs[m_], P::obs_width[m_], P::obs_history[m_], P::obs_items[m_], [&](auto& item_body) GF_INLINE_LAMBDA {
                static_for<P::obs_items[m_]>([&](auto I) GF_INLINE_LAMBDA {
                    constexpr int i_ = decltype(I)::value;
                    GfObsItem it = karg.obs[m_].items[i_];
                    constexpr ItemSig sig = P::item[m_][i_];
                    it.op = sig.op; it.width = sig.width; it.i0 = sig.i0;
                    item_body(it, sig.scaled, sig.noisy, ws_obs_norm_slot<P>(m_, i_));
                });
            });
        });
    } else {
        for (int m = 0; m < n_obs; ++m) {
            const PostObs& ob = a.obs[m];
            obs_manager(m, ob, uni(ob.width), uni(ob.history), uni(ob.num_items), [&](auto& item_body) GF_INLINE_LAMBDA {
                const int n_items = uni(ob.num_items);
                for (int i = 0; i < n_items; ++i) {
                    GfObsItem it = ob.items[i];  // by value: the tile stores must not force reloads of the item
                    it.op = uni(it.op); it.width = uni(it.width); it.scale = uni(it.scale); it.noise = uni(it.noise);
                    item_body(it, it.scale != 1.0f, it.noise != 0.0f, 0);
                }
            });
        }
    }
    if (n_obs == 0) __syncthreads();

    // ---- state the reward wave may still have been reading straight from memory while the tile was assembled: new commands and
    //      the air-time state of reset envs are stored only now, behind the barrier every wave passed after its own work
    if (wave == 0) {
        const int dirty = (int)xch[X_DIRTY * kEnvBlock + lane];
#pragma unroll
        for (int c = 0; c < GF_POST_MAX_CMD; ++c) {
            if (c < n_cmd && (dirty & (1 << c)) && live) {
                const PostCmd cm = cmd_row(c);
                GF_GLOBAL float* crow = G(cm.command) + n * cm.width;
#pragma unroll
                for (int j = 0; j < kPostMaxRanges; ++j)
                    if (j < cm.width) crow[j] = xch[(X_CMD + c * kPostMaxRanges + j) * kEnvBlock + lane];
            }
        }
        if (has_gait && live) {   // the reward wave read the PRE-step row straight from memory (gait_phase / foot_height terms)
            GF_GLOBAL f32x4* grw = reinterpret_cast<GF_GLOBAL f32x4*>(G(UNI(a.gait.state)) + n * GF_GAIT_ROW);
            grw[0] = f32x4{grow[0], grow[1], grow[2], grow[3]};
            grw[1] = f32x4{grow[4], grow[5], grow[6], grow[7]};
            grw[2] = f32x4{grow[8], grow[9], grow[10], grow[11]};
            grw[3] = f32x4{grow[12], grow[13], grow[14], grow[15]};
            if (gait_resampled) G(UNI(a.gait.selected))[n] = (int64_t)gait_sel;
        }
        if (done) {
            const int n_air = UNI(a.n_air);
            for (int m = 0; m < n_air; ++m) {
                const int L = a.air_links[m];
                for (int s = 0; s < 4; ++s) {
                    GF_GLOBAL float* p = G(a.air_state[m][s]);
                    if (p)
                        for (int l = 0; l < L; ++l) p[n * L + l] = 0.0f;
                }
            }
        }
    }
    GF_WSTAMP(8);
}

}  // namespace gf
