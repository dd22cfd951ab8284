// gf_reset.hip — Phase R: masked reset of every piece of manager-owned state, one launch.
//
// Replaces the fan-out of ManagedEnvironment.reset (managed_env.py:336-366):
//   GenesisEnv.reset              genesis_env.py:233-252   actions/last_actions/episode_length <- 0, max length jitter
//   PositionActionManager.reset   position_action_manager.py:455-464   dof_pos <- default (+noise)   [scene side]
//   EntityManager.reset → mdp.reset.position   mdp/reset.py:102-124   pos/quat <- fixed pose, zero velocity [scene side]
//   ContactManager.reset          contact_manager.py:316-329   4 air-time arrays <- 0
//   RewardManager.reset           reward_manager.py:202-222    value/=secs; mean -> log; value<-0; secs<-1e-10
// The reference reaches these through nonzero() (host sync) + index gathers/scatters (~84 aten ops,
// 6 .item() syncs).  Here the done mask is applied per lane; waves without a done env exit after
// two byte loads.  The per-term episode means are wave-reduced in f64 and added with one f64
// atomic per wave and term (the only floating atomics in the library: Σ over ≤N values feeding a
// log line, accumulated in double so the f32-rounded result does not depend on arrival order).
#include "gf_launch.h"

namespace gf {

__device__ __forceinline__ void reset_body(const GfResetArgs& a) {
    const int lane = (int)(threadIdx.x & (GF_WAVE - 1));   // (one wave per 64 envs: whichever wave of a wider workgroup runs the body)
    const int64_t n = (int64_t)blockIdx.x * kEnvBlock + lane;
    const bool live = n < a.num_envs;
    const bool go = live && (a.mask[n] || (a.mask2 && a.mask2[n]));
    const unsigned long long wave_go = __ballot(go);
    if (!wave_go) return;  // wave-uniform
    const int64_t N = a.num_envs;
    const int D = a.num_dofs;

    if (a.stats && lane == 0) atomicAdd(&stats_shard(a.stats)->reset_count, popc64(wave_go));

    // ---- RewardManager.reset: needs every lane for the wave reduction --------------------------
    if (a.episode_seconds) {
        if (a.reward_logging && a.episode_sums) {
            const float secs = go ? a.episode_seconds[n] : 1.0f;
            // every row first: the loads are independent of each other, but each would wait behind the previous term's
            // atomic and store if it were issued inside the loop below (one memory latency per term).  Staged through LDS
            // (lane-private slots) so that loop can keep its run-time trip count.
            __shared__ float s_sum[GF_MAX_TERMS][kEnvBlock];
            {
                float sum[GF_MAX_TERMS];
#pragma unroll
                for (int t = 0; t < GF_MAX_TERMS; ++t) sum[t] = (go && t < a.num_reward_terms) ? a.episode_sums[(int64_t)t * N + n] : 0.0f;
#pragma unroll
                for (int t = 0; t < GF_MAX_TERMS; ++t) s_sum[t][lane] = sum[t];
            }
            for (int t = 0; t < a.num_reward_terms; ++t) {
                float* v = a.episode_sums + (int64_t)t * N + (live ? n : 0);
                if (a.reward_log_mask & (1u << t)) {
                    const float per_sec = go ? (s_sum[t][lane] / secs) : 0.0f;  // value[envs_idx] /= episode_seconds
                    if (a.stats) {
                        const double s = wave_sum((double)per_sec);
                        if (lane == 0) unsafeAtomicAdd(&stats_shard(a.stats)->reward_episode_sum[t], s);  // native global_atomic_add_f64
                    }
                }
                if (go) *v = 0.0f;
            }
        }
        if (go) a.episode_seconds[n] = 1e-10f;
    }
    if (!go) return;

    // ---- GenesisEnv.reset -----------------------------------------------------------------------
    if (a.env_actions) {
        float* r0 = a.env_actions + n * D;
        float* r1 = a.env_last_actions + n * D;
        for (int d = 0; d < D; ++d) { r0[d] = 0.0f; r1[d] = 0.0f; }
    }
    if (a.episode_length) a.episode_length[n] = 0;
    if (a.max_episode_length && a.max_random_scaling > 0.0f) {
        const float u = draw_u(a.len_draws, n, a.seed, a.stream, (uint32_t)n + a.env_offset, 0u);
        const float r = uniform_range(u, -1.0f, 1.0f) * a.max_random_scaling;
        a.max_episode_length[n] = (int32_t)rintf((float)a.base_max_episode_length + r);  // torch.round: half-to-even
    }

    // ---- ContactManager.reset -------------------------------------------------------------------
    for (int m = 0; m < a.num_contact; ++m) {
        const int L = a.air_links[m];
        for (int s = 0; s < 4; ++s) {
            float* p = a.air_state[m][s];
            if (p)
                for (int l = 0; l < L; ++l) p[n * L + l] = 0.0f;
        }
    }

    // ---- scene side (synthetic scene exposes masked setters; NULL for real Genesis) ----------------
    if (a.scene_dof_pos) {
        // four DOFs per pass: the draws of columns 4 + d0 … 4 + d0 + 3 are the four words of ONE Philox block (block 1 + d0 / 4) — a
        // block per DOF computed the same ten rounds four times over (12 blocks instead of 3 for a Go2: most of what a wave with a
        // done env spent here)
        const uint32_t genv = (uint32_t)n + a.env_offset;
        for (int d0 = 0; d0 < D; d0 += 4) {
            float u[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.dof_noise_scale != 0.0f) {
                if (a.dof_draws) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = a.dof_draws[n * D + (d0 + j < D ? d0 + j : D - 1)];
                } else {
                    const U4 r = philox4x32_10(genv, (uint32_t)(4 + d0) >> 2, (uint32_t)a.stream, (uint32_t)(a.stream >> 32), (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
                    u[0] = u24_to_unit(r.x); u[1] = u24_to_unit(r.y); u[2] = u24_to_unit(r.z); u[3] = u24_to_unit(r.w);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = d0 + j;
                if (d < D) {
                    float p = a.default_dof_pos[d];
                    if (a.dof_noise_scale != 0.0f) p = p + uniform_range(u[j], -1.0f, 1.0f) * a.dof_noise_scale;
                    a.scene_dof_pos[n * D + d] = p;
                    if (a.scene_dof_vel) a.scene_dof_vel[n * D + d] = 0.0f;
                }
            }
        }
    }
    if (a.scene_pos) {
        float p[3] = {a.reset_pos[0], a.reset_pos[1], a.reset_pos[2]};
        float4 q = make_float4(a.reset_quat[0], a.reset_quat[1], a.reset_quat[2], a.reset_quat[3]);
        bool set_quat = a.set_quat != 0;
        if (a.spawn_mode) {  // mdp.reset.randomize_terrain_position
            float u[5];
            spawn_draws(a.spawn_draws, n, a.seed, a.stream, (uint32_t)n + a.env_offset, a.spawn_rot_mask, u);
            spawn_pose(a, u, p, &q);
            set_quat = a.spawn_set_quat != 0;
        }
        for (int j = 0; j < 3; ++j) a.scene_pos[3 * n + j] = p[j];
        if (set_quat && a.scene_quat) {
            float4* qp = reinterpret_cast<float4*>(a.scene_quat) + n;
            if (a.quat_stash) reinterpret_cast<float4*>(a.quat_stash)[n] = *qp;  // pre-reset quat for this tick's observation
            *qp = q;
        }
        if (a.zero_velocity) {
            if (a.scene_lin_vel) for (int j = 0; j < 3; ++j) a.scene_lin_vel[3 * n + j] = 0.0f;
            if (a.scene_ang_vel) for (int j = 0; j < 3; ++j) a.scene_ang_vel[3 * n + j] = 0.0f;
            if (a.scene_dof_vel) for (int d = 0; d < D; ++d) a.scene_dof_vel[n * D + d] = 0.0f;
        }
    }
}

#ifndef GF_BODIES_ONLY
__global__ __launch_bounds__(kEnvBlock) void reset_kernel(const GfResetArgs a) {
    prefetch_args<GfResetArgs>();
    reset_body(a);
}
#endif

}  // namespace gf

#ifndef GF_BODIES_ONLY
namespace gf {

// ---- gf_done_compact: the ascending index list of the done envs --------------------------------------------------------------
// A workgroup of 256 lanes owns 4 096 envs, a lane 16 consecutive mask bytes (one dwordx4 per mask when the rows are 16-byte
// aligned).  Above 131 072 envs two launches: launch 1 leaves each block's count; launch 2 adds up the counts in front of its block (at most 256
// of them at 1 M envs), scans its lanes' counts (wave prefix by DPP-free shuffles + 4 wave totals through LDS) and writes the indices in order.
constexpr int kCompactBlock = 256, kCompactPerLane = 16, kCompactEnvs = kCompactBlock * kCompactPerLane;

__device__ __forceinline__ uint32_t compact_bits(const GfCompactArgs& a, const int64_t first) {
    // bit j: env first + j is listed (first is a multiple of 16)
    uint32_t bits = 0;
    const int64_t N = a.num_envs;
    if (first >= N) return 0u;
    const bool vec = first + kCompactPerLane <= N && ((reinterpret_cast<uintptr_t>(a.mask) | reinterpret_cast<uintptr_t>(a.mask2)) & 15u) == 0;
    if (vec) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 w = *reinterpret_cast<const GF_GLOBAL u32x4*>(G(a.mask) + first);
        if (a.mask2) w |= *reinterpret_cast<const GF_GLOBAL u32x4*>(G(a.mask2) + first);
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b) bits |= ((ws[k] >> (8 * b)) & 0xffu) ? 1u << (4 * k + b) : 0u;
    } else {
        for (int j = 0; j < kCompactPerLane && first + j < N; ++j)
            if (G(a.mask)[first + j] || (a.mask2 && G(a.mask2)[first + j])) bits |= 1u << j;
    }
    return bits;
}

__global__ __launch_bounds__(kCompactBlock) void compact_count_kernel(const GfCompactArgs a) {
    const int64_t first = ((int64_t)blockIdx.x * kCompactBlock + threadIdx.x) * kCompactPerLane;
    const int mine = __builtin_popcount(compact_bits(a, first));
    const int w = (int)wave_sum((double)mine);   // (exact: counts are far below 2^53)
    __shared__ int s_w[kCompactBlock / GF_WAVE];
    if ((threadIdx.x & (GF_WAVE - 1)) == 0) s_w[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int i = 0; i < kCompactBlock / GF_WAVE; ++i) t += s_w[i];
        a.block_counts[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(kCompactBlock) void compact_write_kernel(const GfCompactArgs a, const int num_blocks) {
    __shared__ int s_base, s_w[kCompactBlock / GF_WAVE];
    // the counts in front of this block (every lane sums a strided share; num_blocks <= 256 at 1 M envs)
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += kCompactBlock) part += a.block_counts[b];
    const int wpart = (int)wave_sum((double)part);
    if ((threadIdx.x & (GF_WAVE - 1)) == 0) s_w[threadIdx.x >> 6] = wpart;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int i = 0; i < kCompactBlock / GF_WAVE; ++i) t += s_w[i];
        s_base = t;
    }
    __syncthreads();
    const int base = s_base;
    const int64_t first = ((int64_t)blockIdx.x * kCompactBlock + threadIdx.x) * kCompactPerLane;
    uint32_t bits = compact_bits(a, first);
    const int mine = __builtin_popcount(bits);
    // exclusive prefix inside the wave, then across the four waves
    const int lane = threadIdx.x & (GF_WAVE - 1);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < GF_WAVE; o <<= 1) {
        const int up = __shfl_up(incl, o, GF_WAVE);
        if (lane >= o) incl += up;
    }
    __syncthreads();   // (s_w is reused)
    if (lane == GF_WAVE - 1) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    int off = base + incl - mine;
    for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) off += s_w[i];
    while (bits) {
        const int j = __builtin_ctz(bits);
        bits &= bits - 1u;
        G(a.ids_out)[off++] = first + j;
    }
    if (blockIdx.x == (unsigned)num_blocks - 1 && threadIdx.x == kCompactBlock - 1) *a.count_out = off;   // the last lane of the last block ends at the total
}

// Up to kCompactSingle blocks (131 072 envs): ONE launch.  A block finds the count in front of it by counting the masks of the blocks
// before it itself — at most 31 x 4 KiB of cache-resident bytes per block, read as 16-byte units by its 256 lanes — instead of
// waiting for a launch that left per-block counts: a launch less on a path whose launches are all a few microseconds of latency.
constexpr int kCompactSingle = 32;

__global__ __launch_bounds__(kCompactBlock) void compact_single_kernel(const GfCompactArgs a, const int num_blocks) {
    __shared__ int s_base, s_w[kCompactBlock / GF_WAVE];
    const int64_t first = ((int64_t)blockIdx.x * kCompactBlock + threadIdx.x) * kCompactPerLane;
    uint32_t bits = compact_bits(a, first);   // (requested first: the units in front need no result of it)
    int part = 0;
    // The masks in front of this block: whole blocks of kCompactBlock 16-env units, so every lane runs the same blockIdx.x passes —
    // eight at a time, as plain 16-byte loads issued together (compact_bits() has a branch per call: one pass at a time was a memory
    // round trip per block in front, 6 of the kernel's 11 µs at 65 536 envs).  Unaligned masks take the general path.
    typedef uint32_t u32x4c __attribute__((ext_vector_type(4)));
    auto nonzero_bytes = [](u32x4c w) __attribute__((always_inline)) {
        int n = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t x = w[k];
            n += __builtin_popcount((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u);   // one bit per non-zero byte
        }
        return n;
    };
    if (((reinterpret_cast<uintptr_t>(a.mask) | reinterpret_cast<uintptr_t>(a.mask2)) & 15u) == 0) {
        const GF_GLOBAL u32x4c* m1 = reinterpret_cast<const GF_GLOBAL u32x4c*>(G(a.mask));
        const GF_GLOBAL u32x4c* m2 = a.mask2 ? reinterpret_cast<const GF_GLOBAL u32x4c*>(G(a.mask2)) : nullptr;
        constexpr int kBatch = 8;
        for (int b0 = 0; b0 < (int)blockIdx.x; b0 += kBatch) {
            u32x4c w[kBatch], w2[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int64_t u = (int64_t)(b0 + j < (int)blockIdx.x ? b0 + j : b0) * kCompactBlock + threadIdx.x;
                w[j] = m1[u];
                w2[j] = m2 ? m2[u] : u32x4c{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int j = 0; j < kBatch; ++j) part += b0 + j < (int)blockIdx.x ? nonzero_bytes(w[j] | w2[j]) : 0;
        }
    } else {
        for (int b = 0; b < (int)blockIdx.x; ++b) part += __builtin_popcount(compact_bits(a, ((int64_t)b * kCompactBlock + threadIdx.x) * kCompactPerLane));
    }
    const int wpart = (int)wave_sum((double)part);
    if ((threadIdx.x & (GF_WAVE - 1)) == 0) s_w[threadIdx.x >> 6] = wpart;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int i = 0; i < kCompactBlock / GF_WAVE; ++i) t += s_w[i];
        s_base = t;
    }
    __syncthreads();
    const int base = s_base;
    const int mine = __builtin_popcount(bits);
    const int lane = threadIdx.x & (GF_WAVE - 1);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < GF_WAVE; o <<= 1) {
        const int up = __shfl_up(incl, o, GF_WAVE);
        if (lane >= o) incl += up;
    }
    __syncthreads();   // (s_w is reused)
    if (lane == GF_WAVE - 1) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    int off = base + incl - mine;
    for (int i = 0; i < (int)(threadIdx.x >> 6); ++i) off += s_w[i];
    while (bits) {
        const int j = __builtin_ctz(bits);
        bits &= bits - 1u;
        G(a.ids_out)[off++] = first + j;
    }
    if (blockIdx.x == (unsigned)num_blocks - 1 && threadIdx.x == kCompactBlock - 1) *a.count_out = off;
}

}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_done_compact(const GfCompactArgs* a, void* stream) {
    if (!a || !a->mask || !a->ids_out || !a->count_out || !a->block_counts) return GF_E_NULL;
    if (a->num_envs < 0 || a->num_envs >= ((int64_t)1 << 31)) return GF_E_RANGE;
    hipStream_t s = (hipStream_t)stream;
    if (a->num_envs == 0) {
        GF_HIP_CHECK(hipMemsetAsync(a->count_out, 0, sizeof(int32_t), s));
        if (a->wait) GF_HIP_CHECK(hipStreamSynchronize(s));
        return GF_OK;
    }
    const int blocks = (int)((a->num_envs + gf::kCompactEnvs - 1) / gf::kCompactEnvs);
    gf::PhaseScope scope(GF_PHASE_COMPACT, s);
    if (blocks <= gf::kCompactSingle) {
        gf::klaunch(gf::compact_single_kernel, dim3(blocks), dim3(gf::kCompactBlock), 0, s, *a, blocks);
    } else {
        gf::klaunch(gf::compact_count_kernel, dim3(blocks), dim3(gf::kCompactBlock), 0, s, *a);
        gf::klaunch(gf::compact_write_kernel, dim3(blocks), dim3(gf::kCompactBlock), 0, s, *a, blocks);
    }
    const int rc = gf::launch_status();
    if (rc != GF_OK || !a->wait) return rc;
    GF_HIP_CHECK(hipStreamSynchronize(s));
    return GF_OK;
}

namespace gf {
int reset_prep(const GfResetArgs* a) {
    if (!a || !a->mask) return GF_E_NULL;
    if (a->num_envs < 0 || a->num_dofs < 0) return GF_E_RANGE;
    if (a->num_reward_terms < 0 || a->num_reward_terms > GF_MAX_TERMS) return GF_E_RANGE;
    if (a->num_contact < 0 || a->num_contact > GF_MAX_CONTACT_VIEWS) return GF_E_RANGE;
    if (a->env_actions && !a->env_last_actions) return GF_E_NULL;
    if (a->scene_dof_pos && !a->default_dof_pos) return GF_E_NULL;
    if (a->spawn_mode && a->terrain.height_field && (a->terrain.rows < 1 || a->terrain.cols < 1)) return GF_E_RANGE;
    return GF_OK;
}
}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_masked_reset(const GfResetArgs* a, void* stream) {
    const int rc = gf::reset_prep(a);
    if (rc) return rc;
    if (a->num_envs == 0) return GF_OK;
    hipStream_t s = (hipStream_t)stream;
    gf::PhaseScope scope(GF_PHASE_RESET, s);
    scope.begin_bracket();
    gf::klaunch(gf::reset_kernel, dim3(gf::env_grid(a->num_envs)), dim3(gf::kEnvBlock), 0, s, *a);
    return gf::launch_status();
}
#endif
