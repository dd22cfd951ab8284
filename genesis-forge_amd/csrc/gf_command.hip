// gf_command.hip — Phase B5: CommandManager.step / reset / resample_command, one launch.
//
// Replaces managers/command/command_manager.py:152-170 and :290-303: the reference builds
// (episode_length % resample_steps == 0).nonzero() (a host sync), then for each range scatters
// buffer.uniform_(lo, hi) into command[ids, i].  Here the predicate is evaluated per lane and the
// draw comes either from a caller-supplied dense U[0,1) array (parity mode) or from Philox keyed
// by (seed, stream, env, range) — same distribution, no compaction, no sync.
// Ranges are passed by value on every call because curricula mutate them (command_manager.py:293-298).
// Algorithmic traffic: R episode_length 4 B/env; on a resample W 4R B (+ R 4R B of draws in parity mode).
#include "gf_launch.h"

namespace gf {

__device__ __forceinline__ void command_body(const GfCommandArgs& a) {
    const int lane = (int)(threadIdx.x & (GF_WAVE - 1));   // (one wave per 64 envs: whichever wave of a wider workgroup runs the body)
    const int64_t n = (int64_t)blockIdx.x * kEnvBlock + lane;
    const bool live = n < a.num_envs;
    bool go = false;
    if (live) {
        if (a.mode == GF_CMD_STEP) go = (a.episode_length[n] % a.resample_steps) == 0;
        else if (a.mode == GF_CMD_MASKED) go = a.mask[n] || (a.mask2 && a.mask2[n]);
        else go = true;
    }
    if (a.stats && a.mode == GF_CMD_STEP) {
        const unsigned long long m = __ballot(go);
        if (m && lane == 0) atomicAdd(&stats_shard(a.stats)->resample_count, popc64(m));
    }
    if (!go) return;
    const int R = a.num_ranges;
    float* row = a.command + n * R;
    for (int i = 0; i < R; ++i) {
        const float u = draw_u(a.draws, n * R + i, a.seed, a.stream, (uint32_t)n + a.env_offset, (uint32_t)i);
        row[i] = uniform_range(u, a.lo[i], a.hi[i]);
    }
}

#ifndef GF_BODIES_ONLY
__global__ __launch_bounds__(kEnvBlock) void command_kernel(const GfCommandArgs a) { command_body(a); }
#endif

}  // namespace gf

#ifndef GF_BODIES_ONLY
namespace gf {
int command_prep(const GfCommandArgs* a) {
    if (!a || !a->command) return GF_E_NULL;
    if (a->num_ranges <= 0 || a->num_ranges > GF_MAX_RANGES || a->num_envs < 0) return GF_E_RANGE;
    if (a->mode == GF_CMD_STEP) {
        if (!a->episode_length) return GF_E_NULL;
        if (a->resample_steps <= 0) return GF_E_RANGE;  // the reference would raise ZeroDivisionError
    } else if (a->mode == GF_CMD_MASKED) {
        if (!a->mask) return GF_E_NULL;
    } else if (a->mode != GF_CMD_ALL) {
        return GF_E_RANGE;
    }
    return GF_OK;
}
}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_command_step(const GfCommandArgs* a, void* stream) {
    const int rc = gf::command_prep(a);
    if (rc) return rc;
    if (a->num_envs == 0) return GF_OK;
    hipStream_t s = (hipStream_t)stream;
    gf::PhaseScope scope(GF_PHASE_COMMAND, s);
    scope.begin_bracket();
    gf::klaunch(gf::command_kernel, dim3(gf::env_grid(a->num_envs)), dim3(gf::kEnvBlock), 0, s, *a);
    return gf::launch_status();
}
#endif
