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
