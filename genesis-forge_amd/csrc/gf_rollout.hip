// gf_rollout.hip — rollout-storage write as one launch (first slice).
//
// Replaces the three `copy_` launches with which the RL library's rollout storage takes over a step's outputs (rsl_rl
// RolloutStorage.add_transitions; call site examples/simple/train.py:125-129): observations[t+1] <- obs, rewards[t] <- reward,
// dones[t] <- terminated | truncated — time-major rows addressed by the caller.  A flat float4 stream over the [N, W]
// observation (every byte read once, written once), the first N/4 lanes also move the reward and fold the two masks.
// When the step's post-physics phases run fused, the same stores are issued by that launch from the tile it holds
// (GfPostRefs.rollout) and this kernel is not launched at all.
// Algorithmic traffic: R 4W + 4 + 2, W 4W + 4 + 1 bytes per env (W = 48: 395 B/env).
#include "gf_launch.h"

namespace gf {

constexpr int kRollBlock = 256;

__global__ __launch_bounds__(kRollBlock) void rollout_kernel(const GfRolloutArgs a, const int64_t total4, const int vec) {
    const int64_t i = (int64_t)blockIdx.x * kRollBlock + threadIdx.x;
    const int64_t N = a.num_envs;
    if (a.obs_out) {
        if (vec == 4) {
            if (i < total4) reinterpret_cast<GF_GLOBAL f32x4*>(G(a.obs_out))[i] = reinterpret_cast<const GF_GLOBAL f32x4*>(G(a.obs))[i];
        } else {
            const int64_t total = N * a.obs_width;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t e = i * 4 + k;
                if (e < total) G(a.obs_out)[e] = G(a.obs)[e];
            }
        }
    }
    if (i < N) {
        if (a.reward_out) G(a.reward_out)[i] = G(a.reward)[i];
        if (a.done_out) G(a.done_out)[i] = (uint8_t)((G(a.terminated)[i] != 0) | (a.truncated && G(a.truncated)[i] != 0));
    }
}

int rollout_prep(const GfRolloutArgs* a) {
    if (!a) return GF_E_NULL;
    if (a->num_envs < 0 || a->obs_width < 0) return GF_E_RANGE;
    if ((a->obs_out && !a->obs) || (a->reward_out && !a->reward) || (a->done_out && !a->terminated)) return GF_E_NULL;
    return GF_OK;
}

}  // namespace gf

namespace gf {

// ---- the policy's rows of a transition (rsl_rl RolloutStorage.add_transitions) + the time-out bootstrap -----------------------
// Flat float4 streams over the three [N, A] arrays (actions, mu, sigma); the first N lanes move the two [N] columns and do
// rewards[t] += gamma * values * time_outs.  R 12A + 8 + 1 (+4), W 12A + 8 (+4) bytes per env.
__global__ __launch_bounds__(kRollBlock) void rollout_policy_kernel(const GfRolloutPolicyArgs a, const int64_t total4, const int vec) {
    const int64_t i = (int64_t)blockIdx.x * kRollBlock + threadIdx.x;
    const int64_t N = a.num_envs, total = N * a.num_actions;
    const float* src[3] = {a.actions, a.mu, a.sigma};
    float* dst[3] = {a.actions_out, a.mu_out, a.sigma_out};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!dst[k]) continue;
        if (vec == 4) {
            if (i < total4) reinterpret_cast<GF_GLOBAL f32x4*>(G(dst[k]))[i] = reinterpret_cast<const GF_GLOBAL f32x4*>(G(src[k]))[i];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t e = i * 4 + j;
                if (e < total) G(dst[k])[e] = G(src[k])[e];
            }
        }
    }
    if (i < N) {
        float v = 0.f;
        if (a.values) v = G(a.values)[i];
        if (a.values_out) G(a.values_out)[i] = v;
        if (a.log_prob_out) G(a.log_prob_out)[i] = G(a.log_prob)[i];
        if (a.time_outs) {   // PPO.process_env_step: rewards += gamma * values * time_outs  (one product, one product, one sum)
            const float to = G(a.time_outs)[i] ? 1.0f : 0.0f;
            G(a.reward_row)[i] = G(a.reward_row)[i] + (a.gamma * v) * to;
        }
    }
}

// ---- GAE (rsl_rl RolloutStorage.compute_returns) -------------------------------------------------------------------------------
// One lane per env, t = T-1 … 0; row t of each time-major array is one coalesced wave access.  All T loads of a lane are
// independent of the recurrence, so they are issued in batches of 12 steps (T = 24: two round trips) ahead of the arithmetic that consumes them.
constexpr int kGaeBatch = 12;
__global__ __launch_bounds__(kRollBlock) void gae_kernel(const GfGaeArgs a) {
    const int64_t n = (int64_t)blockIdx.x * kRollBlock + threadIdx.x;
    const int64_t N = a.num_envs;
    const bool live = n < N;
    const int64_t m = live ? n : N - 1;
    const int T = a.num_steps;
    const float gamma = a.gamma, lam = a.lam;
    float next = G(a.last_values)[m];
    float adv = 0.f;
    double s1 = 0.0, s2 = 0.0;
    for (int t1 = T; t1 > 0; t1 -= kGaeBatch) {
        float r[kGaeBatch], v[kGaeBatch];
        int d[kGaeBatch];
#pragma unroll
        for (int j = 0; j < kGaeBatch; ++j) {
            const int t = t1 - 1 - j;
            const int64_t at = (int64_t)(t >= 0 ? t : 0) * N + m;
            r[j] = G(a.rewards)[at]; v[j] = G(a.values)[at]; d[j] = G(a.dones)[at];
        }
#pragma unroll
        for (int j = 0; j < kGaeBatch; ++j) {
            const int t = t1 - 1 - j;
            if (t < 0) break;
            const float nt = 1.0f - (d[j] ? 1.0f : 0.0f);
            const float delta = (r[j] + (nt * gamma) * next) - v[j];
            adv = delta + ((nt * gamma) * lam) * adv;
            const float ret = adv + v[j];
            const float out = ret - v[j];
            if (live) {
                G(a.returns)[(int64_t)t * N + n] = ret;
                G(a.advantages)[(int64_t)t * N + n] = out;
                s1 += (double)out; s2 += (double)out * (double)out;
            }
            next = v[j];
        }
    }
    if (a.moments) {
        const double w1 = wave_sum(s1), w2 = wave_sum(s2);
        if ((threadIdx.x & (GF_WAVE - 1)) == 0) { unsafeAtomicAdd(&a.moments[0], w1); unsafeAtomicAdd(&a.moments[1], w2); }
    }
}

__global__ __launch_bounds__(kRollBlock) void gae_normalize_kernel(float* adv, const double* moments, const int64_t total) {
    const int64_t i4 = (int64_t)blockIdx.x * kRollBlock + threadIdx.x;
    const double cnt = (double)total, mean = moments[0] / cnt;
    const double var = total > 1 ? (moments[1] - moments[0] * mean) / (cnt - 1.0) : 0.0;   // unbiased, as torch.std
    const float mu = (float)mean, denom = (float)sqrt(var > 0.0 ? var : 0.0) + 1e-8f;
    if (i4 * 4 + 3 < total && (reinterpret_cast<uintptr_t>(adv) & 15u) == 0) {
        f32x4 x = reinterpret_cast<GF_GLOBAL f32x4*>(G(adv))[i4];
        x.x = (x.x - mu) / denom; x.y = (x.y - mu) / denom; x.z = (x.z - mu) / denom; x.w = (x.w - mu) / denom;
        reinterpret_cast<GF_GLOBAL f32x4*>(G(adv))[i4] = x;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t e = i4 * 4 + j;
            if (e < total) G(adv)[e] = (G(adv)[e] - mu) / denom;
        }
    }
}

}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_rollout_policy_write(const GfRolloutPolicyArgs* a, void* stream) {
    if (!a) return GF_E_NULL;
    if (a->num_envs < 0 || a->num_actions < 0) return GF_E_RANGE;
    if ((a->actions_out && !a->actions) || (a->mu_out && !a->mu) || (a->sigma_out && !a->sigma) || (a->values_out && !a->values) ||
        (a->log_prob_out && !a->log_prob) || (a->time_outs && (!a->reward_row || !a->values)))
        return GF_E_NULL;
    if (a->num_envs == 0) return GF_OK;
    const int64_t total = (int64_t)a->num_envs * a->num_actions;
    uintptr_t bits = 0;
    const void* ptrs[] = {a->actions, a->mu, a->sigma, a->actions_out, a->mu_out, a->sigma_out};
    for (const void* p : ptrs) bits |= reinterpret_cast<uintptr_t>(p);
    const bool al = (bits & 15u) == 0 && (total & 3) == 0;
    const int64_t total4 = (total + 3) / 4;
    const int64_t lanes = total4 > a->num_envs ? total4 : a->num_envs;
    hipStream_t s = (hipStream_t)stream;
    gf::PhaseScope scope(GF_PHASE_ROLLOUT_POLICY, s);
    GF_LAUNCH(scope, gf::rollout_policy_kernel, gf::env_grid(lanes, gf::kRollBlock), gf::kRollBlock, 0, s, *a, total4, al ? 4 : 1);
    return gf::launch_status();
}

extern "C" __attribute__((visibility("default"))) int gf_gae(const GfGaeArgs* a, void* stream) {
    if (!a || !a->rewards || !a->values || !a->dones || !a->last_values || !a->returns || !a->advantages) return GF_E_NULL;
    if (a->num_envs < 0 || a->num_steps < 1) return GF_E_RANGE;
    if (a->normalize && !a->moments) return GF_E_NULL;
    if (a->num_envs == 0) return GF_OK;
    hipStream_t s = (hipStream_t)stream;
    if (a->moments) GF_HIP_CHECK(hipMemsetAsync(a->moments, 0, 2 * sizeof(double), s));
    gf::PhaseScope scope(GF_PHASE_GAE, s);
    GF_LAUNCH(scope, gf::gae_kernel, gf::env_grid(a->num_envs, gf::kRollBlock), gf::kRollBlock, 0, s, *a);
    if (a->normalize) {
        const int64_t total = (int64_t)a->num_envs * a->num_steps;
        gf::klaunch(gf::gae_normalize_kernel, dim3(gf::env_grid((total + 3) / 4, gf::kRollBlock)), dim3(gf::kRollBlock), 0, s, a->advantages, (const double*)a->moments, total);
    }
    return gf::launch_status();
}

extern "C" __attribute__((visibility("default"))) int gf_rollout_write(const GfRolloutArgs* a, void* stream) {
    const int rc = gf::rollout_prep(a);
    if (rc) return rc;
    if (a->num_envs == 0) return GF_OK;
    const int64_t total = (int64_t)a->num_envs * a->obs_width;
    const bool al = ((reinterpret_cast<uintptr_t>(a->obs) | reinterpret_cast<uintptr_t>(a->obs_out)) & 15u) == 0 && (total & 3) == 0;
    const int64_t total4 = (total + 3) / 4;
    const int64_t lanes = a->obs_out ? (total4 > a->num_envs ? total4 : a->num_envs) : a->num_envs;
    hipStream_t s = (hipStream_t)stream;
    gf::PhaseScope scope(GF_PHASE_ROLLOUT, s);
    GF_LAUNCH(scope, gf::rollout_kernel, gf::env_grid(lanes, gf::kRollBlock), gf::kRollBlock, 0, s, *a, total4, al ? 4 : 1);
    return gf::launch_status();
}
