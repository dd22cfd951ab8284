// gf_action.hip — Phase A: GenesisEnv.step bookkeeping + PositionActionManager.step, one launch.
//
// Replaces (reference, /root/reference/genesis_forge/):
//   genesis_env.py:196-203          episode_length += 1; last_actions <- actions; actions <- new
//   managers/action/base.py:67-82   raw -> manager copy
//   managers/action/position_action_manager.py:402-414   NaN/Inf scan, a*scale+offset, clamp(lo,hi)
//   managers/action/position_within_limits.py:125-126    clamp(-1,1), a*scale+offset
// which the reference runs as ~13 separate elementwise launches + 2 host syncs.
//
// Layout: the [N,D] arrays are treated as one flat stream of N*D floats; each lane owns one
// float4 (16 B/lane, 1 KiB per wave instruction, fully coalesced).  The [D] constants are
// indexed with (4*i+j) % D and come from L1/K$.  Algorithmic traffic: 20*D B/env
// (R new, R prev, W last, W actions, W targets) + 8 B/env for episode_length.
#include "gf_launch.h"

namespace gf {

__device__ __forceinline__ float action_target(float x, float s, float o, float lo, float hi, int mode) {
    if (mode == GF_ACTION_WITHIN_LIMITS) {
        x = clamp_min(x, -1.0f);
        x = clamp_max(x, 1.0f);
        return x * s + o;
    }
    float t = x * s + o;
    t = clamp_min(t, lo);
    t = clamp_max(t, hi);
    return t;
}

// `upkeep` leading workgroups do nothing but the statistics ring's housekeeping: zero the NEXT step's slot and fold the PREVIOUS
// step's shards into its vector row.  The fold is a chain of two scattered loads and a dozen cross-lane shuffles per entry (≈ 2.5 µs):
// inside the workgroups that also move actions it was those waves' tail, and with it the kernel's (5.8 µs in the benchmark loop at
// 65 536 envs).  On workgroups of their own it runs beside the main work.  `upkeep` is a multiple of 8, so workgroup b + upkeep still
// lands on the XCD workgroup b of the scene / post-physics kernels lands on (round-robin placement, see gf_action_step).
constexpr int kActionUpkeepBlocks = 24;

template <bool VEC4, bool CONST4 = false>
__global__ __launch_bounds__(256) void action_kernel(const GfActionArgs a, const int64_t total, const int upkeep) {
    if ((int)blockIdx.x < upkeep) {
        const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x), nt = (int)(upkeep * blockDim.x);
        if (a.stats_zero) {   // nobody else touches the next slot during this step
            constexpr int kWords = (int)(sizeof(GfStepStats) * GF_STATS_SHARDS / 4);
            for (int w = t; w < kWords; w += nt) reinterpret_cast<uint32_t*>(a.stats_zero)[w] = 0u;
        }
        if (a.stats_fold_src && a.stats_fold_dst) {   // the previous slot is complete by stream order: one entry per wave
            for (int v = t / GF_WAVE; v < GF_STATS_VECTOR_LEN; v += nt / GF_WAVE) fold_stats_entry(a.stats_fold_src, a.stats_fold_dst, a.stats_last_reset, v);
        }
        return;
    }
    const int64_t i = (int64_t)(blockIdx.x - upkeep) * blockDim.x + threadIdx.x;
    const int D = a.num_dofs;
    const int mode = a.mode;
    int flags = 0;

    // episode_length += 1  (genesis_env.py:197) — first N/4 lanes, int4 RMW
    if (a.episode_length) {
        const int64_t N = a.num_envs;
        if (VEC4 && (N & 3) == 0) {
            if (i < (N >> 2)) {
                int4* p = reinterpret_cast<int4*>(a.episode_length) + i;
                int4 v = *p;
                v.x += 1; v.y += 1; v.z += 1; v.w += 1;
                *p = v;
            }
        } else if (i < N) {
            a.episode_length[i] += 1;
        }
    }

    if (VEC4) {
        if (i < (total >> 2)) {
            const float4 x = reinterpret_cast<const float4*>(a.actions_in)[i];
            if (a.env_actions) {
                const float4 prev = reinterpret_cast<const float4*>(a.env_actions)[i];
                reinterpret_cast<float4*>(a.env_last_actions)[i] = prev;
                reinterpret_cast<float4*>(a.env_actions)[i] = x;
            }
            const float xs[4] = {x.x, x.y, x.z, x.w};
            float ts[4];
            const int d0 = (int)((i * 4) % D);
            if (CONST4) {
                // D % 4 == 0: the lane's four elements are four consecutive DOFs that never wrap — one 16-byte load per
                // constant array (L1 / K$ resident) instead of sixteen scalar gathers
                const float4 sc = *reinterpret_cast<const float4*>(a.scale + d0);
                const float4 of = *reinterpret_cast<const float4*>(a.offset + d0);
                float4 lo4 = make_float4(0.f, 0.f, 0.f, 0.f), hi4 = lo4;
                if (mode == GF_ACTION_POSITION) {
                    lo4 = *reinterpret_cast<const float4*>(a.clip_lo + d0);
                    hi4 = *reinterpret_cast<const float4*>(a.clip_hi + d0);
                }
                const float ss[4] = {sc.x, sc.y, sc.z, sc.w}, os[4] = {of.x, of.y, of.z, of.w};
                const float ls[4] = {lo4.x, lo4.y, lo4.z, lo4.w}, hs[4] = {hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ts[j] = action_target(xs[j], ss[j], os[j], ls[j], hs[j], mode);
                    if (a.check_finite && mode == GF_ACTION_POSITION) {
                        flags |= isnan(xs[j]) ? 1 : 0;
                        flags |= isinf(xs[j]) ? 2 : 0;
                    }
                }
            } else {
                int d = d0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float lo = mode == GF_ACTION_POSITION ? a.clip_lo[d] : 0.f;
                    const float hi = mode == GF_ACTION_POSITION ? a.clip_hi[d] : 0.f;
                    ts[j] = action_target(xs[j], a.scale[d], a.offset[d], lo, hi, mode);
                    if (a.check_finite && mode == GF_ACTION_POSITION) {
                        flags |= isnan(xs[j]) ? 1 : 0;
                        flags |= isinf(xs[j]) ? 2 : 0;
                    }
                    d = d + 1 == D ? 0 : d + 1;
                }
            }
            reinterpret_cast<float4*>(a.targets)[i] = make_float4(ts[0], ts[1], ts[2], ts[3]);
        }
    } else {
        if (i < total) {
            const float x = a.actions_in[i];
            if (a.env_actions) {
                a.env_last_actions[i] = a.env_actions[i];
                a.env_actions[i] = x;
            }
            const int d = (int)(i % D);
            const float lo = mode == GF_ACTION_POSITION ? a.clip_lo[d] : 0.f;
            const float hi = mode == GF_ACTION_POSITION ? a.clip_hi[d] : 0.f;
            a.targets[i] = action_target(x, a.scale[d], a.offset[d], lo, hi, mode);
            if (a.check_finite && mode == GF_ACTION_POSITION) {
                flags |= isnan(x) ? 1 : 0;
                flags |= isinf(x) ? 2 : 0;
            }
        }
    }

    // NaN/Inf detection (position_action_manager.py:402-406): the reference syncs twice per step
    // to print; here a flag word is OR-ed on device and polled lazily by the host.
    if (a.stats && a.check_finite) {
        const unsigned long long nan_m = __ballot(flags & 1);
        const unsigned long long inf_m = __ballot(flags & 2);
        if ((nan_m | inf_m) && (threadIdx.x & (GF_WAVE - 1)) == 0) atomicOr(&stats_shard(a.stats)->action_flags, (nan_m ? 1 : 0) | (inf_m ? 2 : 0));
    }
}

}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_action_step(const GfActionArgs* a, void* stream) {
    if (!a || !a->actions_in || !a->targets || !a->scale || !a->offset) return GF_E_NULL;
    if (a->mode == GF_ACTION_POSITION && (!a->clip_lo || !a->clip_hi)) return GF_E_NULL;
    if (a->mode != GF_ACTION_POSITION && a->mode != GF_ACTION_WITHIN_LIMITS) return GF_E_RANGE;
    if (a->env_actions && !a->env_last_actions) return GF_E_NULL;
    if (a->num_envs < 0 || a->num_dofs <= 0) return GF_E_RANGE;
    if (a->num_envs == 0) return GF_OK;
    const int64_t total = (int64_t)a->num_envs * a->num_dofs;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool vec = (total & 3) == 0 && al16(a->actions_in) && al16(a->targets) && (!a->env_actions || (al16(a->env_actions) && al16(a->env_last_actions))) &&
                     (!a->episode_length || al16(a->episode_length));
    hipStream_t s = (hipStream_t)stream;
    gf::PhaseScope scope(GF_PHASE_ACTION, s);
    scope.begin_bracket();
    const int upkeep = (a->stats_zero || (a->stats_fold_src && a->stats_fold_dst)) ? gf::kActionUpkeepBlocks : 0;
    if (vec) {
        int64_t lanes = total >> 2;
        if (a->episode_length) {
            const int64_t need = (a->num_envs & 3) == 0 ? (a->num_envs >> 2) : a->num_envs;
            if (need > lanes) lanes = need;
        }
        const bool const4 = (a->num_dofs & 3) == 0 && al16(a->scale) && al16(a->offset) &&
                            (a->mode != GF_ACTION_POSITION || (al16(a->clip_lo) && al16(a->clip_hi)));
        // D = 12: 192 lanes = the float4s of exactly 64 envs, so workgroup b owns envs [64b, 64b+64) like workgroup b of the scene and
        // post-physics kernels does — with round-robin workgroup → XCD placement a tile stays on one XCD (one L2) across the step
        // (measured in the benchmark loop at 65 536 envs: action kernel 8.6 → 7.1 µs, step 22.8 → 20.4 µs; GF_ACTION_BLOCK256=1 restores
        // the flat 256-lane mapping for comparison)
        static const bool flat256 = getenv("GF_ACTION_BLOCK256") != nullptr;
        static const int forced = getenv("GF_ACTION_BLOCK") ? atoi(getenv("GF_ACTION_BLOCK")) : 0;   // experiments only
        const int block = forced > 0 ? forced : ((a->num_dofs == 12 && !flat256) ? 192 : 256);
        if (const4) gf::klaunch(gf::action_kernel<true, true>, dim3(gf::env_grid(lanes, block) + upkeep), dim3(block), 0, s, *a, total, upkeep);
        else gf::klaunch(gf::action_kernel<true>, dim3(gf::env_grid(lanes, block) + upkeep), dim3(block), 0, s, *a, total, upkeep);
    } else {
        int64_t lanes = total > a->num_envs ? total : a->num_envs;
        gf::klaunch(gf::action_kernel<false>, dim3(gf::env_grid(lanes, 256) + upkeep), dim3(256), 0, s, *a, total, upkeep);
    }
    return gf::launch_status();
}
