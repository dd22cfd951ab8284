// gf_gait.hip — GaitCommandManager.step / reset / resample_command as one launch.
//
// Replaces the user-level manager of the reference's gait_trainer example
// (examples/gait_trainer/gait_command_manager.py): per step it runs the base class's
// (episode_length % resample_steps == 0).nonzero() host sync (command_manager.py:152-162), a torch.multinomial and per-gait
// masked scatters with `mask.any()` syncs (:185-211, 347-399), four `(gait_selected == i).sum()` reductions for the log
// (:430-441), and ≈ 30 small elementwise launches for the clock (:231-239).  Here one lane carries one env's 64-byte state
// row (4 x dwordx4 in, 4 x dwordx4 out, one wave = 4 KiB contiguous): resample predicate, gait selection by inverse CDF,
// the two uniform draws, the per-gait env counts (wave ballots -> one integer atomic per gait per wave into this
// workgroup's statistics shard) and the phase clock.
//
// Arithmetic follows the reference's torch expressions one rounding per op: `%` is aten's remainder (fmod +
// sign fix, exact), `2 * torch.pi * x` multiplies by (float)(2π); sin/cos are the fixed-sequence sincos_det shared with the
// oracle (within 2 ulp of libm; observations are compared at 1e-5).
// Algorithmic traffic: R 64 + W 64 B/env state, R 4 B episode_length, RW 8 B selected only for resampled envs.
#include "gf_launch.h"

namespace gf {

// `flags_all` (masked launches only): rewrite the swing / stance byte of EVERY block, not only of the blocks that hold a reset
// env — the phase chains use it to keep the flags' only writer in a different launch from their reader (gf_chain.hip).
__device__ __forceinline__ void gait_body(const GfGaitArgs& a, const bool flags_all = false) {
    const int lane = (int)(threadIdx.x & (GF_WAVE - 1));   // (one wave per 64 envs: whichever wave of a wider workgroup runs the body)
    const int64_t n = (int64_t)blockIdx.x * kEnvBlock + lane;
    const bool live = n < a.num_envs;
    const int64_t m = live ? n : (int64_t)a.num_envs - 1;  // tail lanes shadow the last env and never store
    const bool step = a.mode == GF_CMD_STEP;
    bool go = false;
    if (live) {
        if (step) go = (a.episode_length[m] % a.resample_steps) == 0;
        else if (a.mode == GF_CMD_MASKED) go = a.mask[m] || (a.mask2 && a.mask2[m]);
        else go = true;
    }
    // a masked launch touches only the 64-env blocks that contain a reset env (resets are sparse): 2 B/env for the rest
    const bool rows = step || __ballot(go) != 0ull;  // wave-uniform
    const bool flags = a.wave_flags != nullptr && (rows || flags_all);
    if (!rows && !flags) return;
    const GF_GLOBAL float* row = G(a.state) + m * GF_GAIT_ROW;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 r0 = ldg4(row), r1 = rows ? ldg4(row + 4) : z4, r2 = rows ? ldg4(row + 8) : z4, r3 = ldg4(row + 12);  // flags need offsets + phase only
    int sel = (step || go) ? (int)G(a.selected)[m] : 0;
    float off[4] = {r0.x, r0.y, r0.z, r0.w};
    float height = r1.x, period = r1.y;
    float clock[8] = {r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y};
    float gtime = r3.z, phase = r3.w;

    if (go) {  // resample_command → _set_gait (:185-211, 347-377)
        float u0, u1, u2;
        if (a.draws) {
            u0 = a.draws[m * 3]; u1 = a.draws[m * 3 + 1]; u2 = a.draws[m * 3 + 2];
        } else {
            const U4 r = philox4x32_10((uint32_t)m + a.env_offset, 0u, (uint32_t)a.stream, (uint32_t)(a.stream >> 32), (uint32_t)a.seed,
                                       (uint32_t)(a.seed >> 32));
            u0 = u24_to_unit(r.x); u1 = u24_to_unit(r.y); u2 = u24_to_unit(r.z);
        }
        int g = 0;
        for (int k = 0; k + 1 < a.num_gaits; ++k) g += (u0 >= a.cum_weight[k]) ? 1 : 0;  // inverse CDF of torch.multinomial's weights
        sel = g;
#pragma unroll
        for (int f = 0; f < 4; ++f) off[f] = a.gait_offsets[g][f];
        height = ((a.fixed_clearance_mask >> g) & 1) ? a.clearance_lo : uniform_range(u1, a.clearance_lo, a.clearance_hi);
        period = uniform_range(u2, a.period_lo, a.period_hi);
        if (!step) {  // reset (:241-255)
#pragma unroll
            for (int j = 0; j < 8; ++j) clock[j] = 0.0f;
            gtime = 0.0f;
            phase = 0.0f;
        }
    }
    if (step) {
        if (a.stats) {  // _log_metrics (:430-441): envs per gait, after this step's resample
#pragma unroll
            for (int g = 0; g < GF_MAX_GAITS; ++g) {
                const unsigned long long b = __ballot(live && sel == g);
                if (b && lane == 0) atomicAdd(&stats_shard(a.stats)->gait_count[g], popc64(b));
            }
        }
        // the periodic clock (:231-239), for every env
        gtime = torch_remainder(gtime + a.dt, period);
        phase = gtime / period;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float fp = torch_remainder_one(phase + off[f]);
            sincos_det(a.two_pi * fp, &clock[f], &clock[4 + f]);
        }
    }
    if (flags) {  // this block's "any env in swing / stance" byte, from the rows as they are after this launch
        const float pi = 0.5f * a.two_pi;  // exact halving: (float)(2π)/2 == (float)π
        uint32_t byte = 0;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int fl = gait_foot_flags(phase, off[f], a.two_pi, pi);
            if (__ballot(live && (fl & 1))) byte |= 1u << (2 * f);
            if (__ballot(live && (fl & 2))) byte |= 2u << (2 * f);
        }
        if (lane == 0) a.wave_flags[blockIdx.x] = (uint8_t)byte;
    }
    if (live && (step || go)) {
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        GF_GLOBAL f32x4v* out = reinterpret_cast<GF_GLOBAL f32x4v*>(G(a.state) + n * GF_GAIT_ROW);
        out[0] = f32x4v{off[0], off[1], off[2], off[3]};
        out[1] = f32x4v{height, period, clock[0], clock[1]};
        out[2] = f32x4v{clock[2], clock[3], clock[4], clock[5]};
        out[3] = f32x4v{clock[6], clock[7], gtime, phase};
        if (go) G(a.selected)[n] = (int64_t)sel;
    }
}

#ifndef GF_BODIES_ONLY
__global__ __launch_bounds__(kEnvBlock) void gait_kernel(const GfGaitArgs a, const int flags_all) { gait_body(a, flags_all != 0); }
#endif

}  // namespace gf

#ifndef GF_BODIES_ONLY
namespace gf {
int gait_prep(const GfGaitArgs* a) {
    if (!a || !a->state || !a->selected) return GF_E_NULL;
    if (a->num_envs < 0 || a->num_gaits < 1 || a->num_gaits > GF_MAX_GAITS) return GF_E_RANGE;
    if (a->mode == GF_CMD_STEP) {
        if (!a->episode_length) return GF_E_NULL;
        if (a->resample_steps <= 0) return GF_E_RANGE;
    } else if (a->mode == GF_CMD_MASKED) {
        if (!a->mask) return GF_E_NULL;
    } else if (a->mode != GF_CMD_ALL) {
        return GF_E_RANGE;
    }
    if (reinterpret_cast<uintptr_t>(a->state) & 15u) return GF_E_UNSUPPORTED;
    return GF_OK;
}
}  // namespace gf

namespace gf {
int gait_launch(const GfGaitArgs* a, hipStream_t s, bool flags_all) {
    const int rc = gait_prep(a);
    if (rc) return rc;
    if (a->num_envs == 0) return GF_OK;
    PhaseScope scope(GF_PHASE_GAIT, s);
    GF_LAUNCH(scope, gait_kernel, env_grid(a->num_envs), kEnvBlock, 0, s, *a, (int)(flags_all && a->mode != GF_CMD_STEP));
    return launch_status();
}
}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_gait_step(const GfGaitArgs* a, void* stream) {
    return gf::gait_launch(a, (hipStream_t)stream, false);
}
#endif
