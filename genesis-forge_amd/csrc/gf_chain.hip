// gf_chain.hip — phase chains: consecutive per-env phases of a recorded step in ONE launch.
//
// A config the fused post-physics kernel (gf_post.hip) does not cover — the gait config with its second kind of command
// manager, a robot whose DOF count has no fused variant, … — replays its post-physics phases one by one: termination, reward,
// command.step ×k, reset, command.reset ×k, observe ×m = 9 launches for the gait config.  Below ≈ 32 k envs a step is bound by
// launch count (≈ 3.5 µs of host time and ≈ 5 µs of GPU latency per launch), not by bytes.  Every one of those
// phases maps workgroup b to the 64 envs [64b, 64b+64) and a later phase reads, for an env, only what an earlier phase wrote
// for the SAME env — so they can run back to back inside one workgroup, separated by a workgroup barrier (which makes the
// earlier phase's global stores visible to the whole workgroup: all its waves share one L1):
//   chain A  (64 lanes / workgroup):   termination → reward → command.step / gait.step of every command manager
//   chain B  (256 lanes / workgroup):  masked reset → command.reset / gait.reset (wave 0) → every ObservationManager (4 waves)
// One value crosses workgroups: the gait manager's per-block swing / stance bytes, which the reward terms of the workgroup
// that owns env 0 read for ALL blocks (GfRewardArgs.gait_wave_flags).  Inside chain A that reader would share a launch with
// gait.step, the bytes' writer, in other workgroups.  So chain A folds a gait.step op only when a masked gait launch over the
// same state follows in the op list: gait.step then leaves the bytes alone and that later launch (chain B or stand-alone)
// rewrites every block's byte from the post-step, post-reset rows — reader and writer never share a launch.
// The phase bodies are the very functions the stand-alone kernels run (gf_*.hip, compiled here with GF_BODIES_ONLY), and the
// descriptors are the ordinary per-phase ones, validated by the same *_prep functions: a chain is by construction "the phases
// in sequence".  gf_run_ops folds a matching run of ops into a chain; anything else launches phase by phase as before.
#define GF_BODIES_ONLY
#include "gf_termination.hip"
#include "gf_reward.hip"
#include "gf_command.hip"
#include "gf_gait.hip"
#include "gf_reset.hip"
#include "gf_observe.hip"
#undef GF_BODIES_ONLY

namespace gf {

int termination_prep(const GfTerminationArgs* a, uint32_t* needs_out);
int reward_prep(const GfRewardArgs* a, uint32_t* needs_out, int* dv_out);
int command_prep(const GfCommandArgs* a);
int gait_prep(const GfGaitArgs* a);
int reset_prep(const GfResetArgs* a);
int observe_prep(const GfObservationArgs* a, uint32_t* needs_out, int* vec_out);

constexpr int kChainCmd = 2, kChainGait = 1, kChainObs = 2;

struct ChainAArgs {
    GfTerminationArgs term;
    GfRewardArgs rew;
    GfCommandArgs cmd[kChainCmd];
    GfGaitArgs gait[kChainGait];
    uint32_t needs_t, needs_r;
    int32_t has_reward, n_cmd, n_gait;
    int32_t has_term;   // 0: the chain starts at the reward op (the termination op ran in an earlier launch: a recorded step cut
                        // in front of a reward manager with Python-level terms)
};
static_assert(sizeof(ChainAArgs) <= 4096, "kernarg segment");

struct ChainBArgs {
    GfResetArgs reset;
    GfCommandArgs cmd[kChainCmd];
    GfGaitArgs gait[kChainGait];
    GfObservationArgs obs[kChainObs];
    uint32_t needs_o[kChainObs];
    int32_t vec[kChainObs];
    int32_t n_cmd, n_gait, n_obs;
    int32_t gait_flags_all;   // bit g: gait[g] rewrites the swing / stance byte of every block (its step ran in chain A)
};
static_assert(sizeof(ChainBArgs) <= 4096, "kernarg segment");

template <int DV>
__global__ __launch_bounds__(kEnvBlock) void chain_a_kernel(const ChainAArgs a) {
    __shared__ float lds_sums[GF_MAX_TERMS * kEnvBlock];
    if (a.has_term) termination_body(a.term, a.needs_t);
    __syncthreads();  // the masks this workgroup just wrote are read by its reward terms
    if (a.has_reward) reward_body<DV>(a.rew, a.needs_r, lds_sums);
    __syncthreads();  // reward terms read the commands BEFORE this step's resample (managed_env.py:312-319)
    // constant indices only: a run-time index into a by-value kernel argument sends the whole struct to scratch memory
#pragma unroll
    for (int c = 0; c < kChainCmd; ++c)
        if (c < a.n_cmd) command_body(a.cmd[c]);
#pragma unroll
    for (int g = 0; g < kChainGait; ++g)
        if (g < a.n_gait) gait_body(a.gait[g]);
}

template <int M>
__device__ __forceinline__ void chain_observe(const ChainBArgs& a, float* tile) {
    if (M >= a.n_obs) return;
    observe_body(a.vec[M], a.obs[M], a.needs_o[M], tile, (uint32_t)(offsetof(ChainBArgs, obs) + M * sizeof(GfObservationArgs)));
}

__global__ __launch_bounds__(kObsBlock) void chain_b_kernel(const ChainBArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    // The per-env phases are one wave per 64 envs each — and independent of each other (the masked command / gait launches read the
    // termination masks and their own state, nothing the reset writes), so three waves run them side by side instead of one wave
    // one after the other: the reset's path for a tile with a done env is a chain of ≈ 6 memory round trips by itself.
    const int wave = (int)(threadIdx.x >> 6);
    if (wave == 0) {
        reset_body(a.reset);
    } else if (wave == 1) {
#pragma unroll
        for (int c = 0; c < kChainCmd; ++c)
            if (c < a.n_cmd) command_body(a.cmd[c]);
    } else if (wave == 2) {
#pragma unroll
        for (int g = 0; g < kChainGait; ++g)
            if (g < a.n_gait) gait_body(a.gait[g], ((a.gait_flags_all >> g) & 1) != 0);
    }
    __syncthreads();  // observations read the post-reset state and the new commands (managed_env.py:322-326)
    chain_observe<0>(a, tile);
    if (a.n_obs > 1) {  // wave-uniform
        __syncthreads();  // the LDS tile is reused
        chain_observe<1>(a, tile);
    }
}

static bool profiled(int phase) { return g_prof.phase == phase; }

// ops[i] is a termination op (or a reward op whose termination op ran earlier): fold it with the reward / command.step /
// gait.step ops that follow.  Returns the number of ops consumed (0 = not a chain: launch phase by phase), *rc = launch status.
int chain_a_try(const GfOp* ops, int i, int num_ops, hipStream_t s, int* rc, DeferredFlags* deferred) {
    if (!g_options[GF_OPT_CHAIN] || profiled(GF_PHASE_TERMINATION) || profiled(GF_PHASE_REWARD) || profiled(GF_PHASE_COMMAND) || profiled(GF_PHASE_GAIT)) return 0;
    ChainAArgs k{};
    const GfTerminationArgs* t = nullptr;
    int j = i, dv = 0, N = 0;
    if (ops[i].phase == GF_PHASE_TERMINATION) {
        t = (const GfTerminationArgs*)ops[i].args;
        if (!t || termination_prep(t, &k.needs_t) != GF_OK || t->num_envs <= 0) return 0;
        N = t->num_envs;
        ++j;
    }
    const GfRewardArgs* r = nullptr;
    if (j < num_ops && ops[j].phase == GF_PHASE_REWARD) {
        r = (const GfRewardArgs*)ops[j].args;
        if (!r || (t && r->num_envs != N) || r->num_envs <= 0 || r->mode != GF_REWARD_MODE_STEP || reward_prep(r, &k.needs_r, &dv) != GF_OK) return 0;
        N = r->num_envs;
        ++j;
    }
    if (!t && !r) return 0;
    for (; j < num_ops; ++j) {
        if (ops[j].phase == GF_PHASE_COMMAND) {
            const GfCommandArgs* c = (const GfCommandArgs*)ops[j].args;
            if (!c || c->mode != GF_CMD_STEP || c->num_envs != N || k.n_cmd >= kChainCmd || command_prep(c) != GF_OK) break;
            k.cmd[k.n_cmd++] = *c;
        } else if (ops[j].phase == GF_PHASE_GAIT) {
            const GfGaitArgs* g = (const GfGaitArgs*)ops[j].args;
            if (!g || g->mode != GF_CMD_STEP || g->num_envs != N || k.n_gait >= kChainGait || gait_prep(g) != GF_OK) break;
            GfGaitArgs& kg = k.gait[k.n_gait];
            kg = *g;
            if (g->wave_flags) {
                // the bytes are read across workgroups by this very launch's reward terms: hand their update to the masked
                // gait launch that follows, or keep gait.step out of the chain
                bool later = false;
                for (int q = j + 1; q < num_ops && !later; ++q)
                    if (ops[q].phase == GF_PHASE_GAIT && ops[q].args) {
                        const GfGaitArgs* m = (const GfGaitArgs*)ops[q].args;
                        later = m->mode != GF_CMD_STEP && m->state == g->state && m->wave_flags == g->wave_flags && m->num_envs == N;
                    }
                if (!later || !deferred->push(g->state)) break;
                kg.wave_flags = nullptr;
            }
            ++k.n_gait;
        } else {
            break;
        }
    }
    if (j - i < 2) return 0;
    k.has_term = t ? 1 : 0;
    if (t) k.term = *t;
    k.has_reward = r ? 1 : 0;
    if (r) k.rew = *r;
    const dim3 grid(env_grid(N)), block(kEnvBlock);
    if (dv == 3) klaunch(chain_a_kernel<3>, grid, block, 0, s, k);
    else if (dv == 7) klaunch(chain_a_kernel<7>, grid, block, 0, s, k);
    else klaunch(chain_a_kernel<0>, grid, block, 0, s, k);
    *rc = launch_status();
    return j - i;
}

// ops[i] is a masked-reset op: fold it with the command.reset / gait.reset and observe ops that follow.
int chain_b_try(const GfOp* ops, int i, int num_ops, hipStream_t s, int* rc, DeferredFlags* deferred) {
    if (!g_options[GF_OPT_CHAIN] || profiled(GF_PHASE_RESET) || profiled(GF_PHASE_COMMAND) || profiled(GF_PHASE_GAIT) || profiled(GF_PHASE_OBSERVE)) return 0;
    ChainBArgs k{};
    const GfResetArgs* r = (const GfResetArgs*)ops[i].args;
    if (!r || reset_prep(r) != GF_OK || r->num_envs <= 0) return 0;
    const int N = r->num_envs;
    int j = i + 1;
    size_t lds = 0;
    for (; j < num_ops; ++j) {
        if (ops[j].phase == GF_PHASE_COMMAND && k.n_obs == 0) {
            const GfCommandArgs* c = (const GfCommandArgs*)ops[j].args;
            if (!c || c->mode != GF_CMD_MASKED || c->num_envs != N || k.n_cmd >= kChainCmd || command_prep(c) != GF_OK) break;
            k.cmd[k.n_cmd++] = *c;
        } else if (ops[j].phase == GF_PHASE_GAIT && k.n_obs == 0) {
            const GfGaitArgs* g = (const GfGaitArgs*)ops[j].args;
            if (!g || g->mode != GF_CMD_MASKED || g->num_envs != N || k.n_gait >= kChainGait || gait_prep(g) != GF_OK) break;
            if (deferred->has(g->state)) k.gait_flags_all |= 1 << k.n_gait;
            k.gait[k.n_gait++] = *g;
        } else if (ops[j].phase == GF_PHASE_OBSERVE) {
            const GfObservationArgs* o = (const GfObservationArgs*)ops[j].args;
            if (!o || o->num_envs != N || k.n_obs >= kChainObs || observe_prep(o, &k.needs_o[k.n_obs], &k.vec[k.n_obs]) != GF_OK) break;
            const size_t need = (size_t)(o->obs_width + 1) * kEnvBlock * sizeof(float);
            lds = need > lds ? need : lds;
            k.obs[k.n_obs++] = *o;
        } else {
            break;
        }
    }
    if (j - i < 2) return 0;
    for (int g = 0; g < k.n_gait; ++g)
        if ((k.gait_flags_all >> g) & 1) deferred->pop(k.gait[g].state);
    k.reset = *r;
    klaunch(chain_b_kernel, dim3(env_grid(N)), dim3(kObsBlock), lds, s, k);
    *rc = launch_status();
    return j - i;
}

}  // namespace gf
