// microbench.hip — standalone timing harness for the phase kernels of libgf_step.so (development tool).
// Builds the Go2 command-config descriptors on hipMalloc'd buffers and times back-to-back launches with
// HIP events; run it under rocprofv3 --kernel-trace --stats for pure kernel durations.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip -Lgenesis-forge_amd -lgf_step -Wl,-rpath,'$ORIGIN/../genesis-forge_amd'
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../include/gf_step.h"

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

template <typename T>
T* dalloc(size_t n, const std::vector<T>* init = nullptr) {
    T* p;
    CK(hipMalloc(&p, n * sizeof(T)));
    if (init) CK(hipMemcpy(p, init->data(), n * sizeof(T), hipMemcpyHostToDevice));
    else CK(hipMemset(p, 0, n * sizeof(T)));
    return p;
}

static std::mt19937 rng(1234);
std::vector<float> randn(size_t n, float mu = 0.f, float sd = 1.f) {
    std::normal_distribution<float> d(mu, sd);
    std::vector<float> v(n);
    for (auto& x : v) x = d(rng);
    return v;
}

__global__ void empty_kernel(int n) {}

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) out[i] = in[i];
}

template <typename F>
double time_loop(const char* name, int iters, double bytes, F&& f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    printf("%-28s %8.2f us/launch  %8.1f GB/s (algorithmic)\n", name, us, bytes / us / 1e3);
    return us;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 65536;
    const int iters = argc > 2 ? atoi(argv[2]) : 500;
    const int D = 12, T = 6, O = 48;
    printf("N=%d D=%d iters=%d\n", N, D, iters);

    auto q = randn((size_t)N * 4, 0.f, 0.02f);  // ~2 deg tilt: a few percent of envs terminate, like the bench
    for (int n = 0; n < N; ++n) {
        q[4 * n] += 1.f;
        float s = 0;
        for (int j = 0; j < 4; ++j) s += q[4 * n + j] * q[4 * n + j];
        s = std::sqrt(s);
        for (int j = 0; j < 4; ++j) q[4 * n + j] /= s;
    }
    auto hpos = randn((size_t)N * 3, 0.f, 0.05f);
    for (int n = 0; n < N; ++n) hpos[3 * n + 2] += 0.35f;
    auto hlin = randn((size_t)N * 3), hang = randn((size_t)N * 3);
    auto hdof = randn((size_t)N * D, 0.f, 0.3f), hvel = randn((size_t)N * D, 0.f, 2.f), hact = randn((size_t)N * D), hlast = randn((size_t)N * D);
    auto hcmd = randn((size_t)N * 3, 0.f, 0.5f);
    std::vector<float> hdef = {0, 0.8f, -1.5f, 0, 0.8f, -1.5f, 0, 1.0f, -1.5f, 0, 1.0f, -1.5f};
    std::vector<float> hscale(D, 0.25f), hlo(D, -3.f), hhi(D, 3.f);
    std::vector<int32_t> hep(N), hmax(N);
    for (int n = 0; n < N; ++n) { hep[n] = rng() % 1100; hmax[n] = 900 + rng() % 200; }

    float *pos = dalloc<float>((size_t)N * 3, &hpos), *quat = dalloc<float>((size_t)N * 4, &q), *lin = dalloc<float>((size_t)N * 3, &hlin),
          *ang = dalloc<float>((size_t)N * 3, &hang), *dof = dalloc<float>((size_t)N * D, &hdof), *dvel = dalloc<float>((size_t)N * D, &hvel),
          *act = dalloc<float>((size_t)N * D, &hact), *last = dalloc<float>((size_t)N * D, &hlast), *cmd = dalloc<float>((size_t)N * 3, &hcmd),
          *def = dalloc<float>(D, &hdef), *scale = dalloc<float>(D, &hscale), *lo = dalloc<float>(D, &hlo), *hi = dalloc<float>(D, &hhi),
          *targets = dalloc<float>((size_t)N * D), *act_in = dalloc<float>((size_t)N * D, &hact), *reward = dalloc<float>(N),
          *secs = dalloc<float>(N), *sums = dalloc<float>((size_t)T * N), *obs = dalloc<float>((size_t)N * O), *stash = dalloc<float>((size_t)N * 4);
    int32_t *ep = dalloc<int32_t>(N, &hep), *maxlen = dalloc<int32_t>(N, &hmax);
    uint8_t *term = dalloc<uint8_t>(N), *trunc = dalloc<uint8_t>(N);
    GfStepStats* stats = dalloc<GfStepStats>(GF_STATS_SHARDS);

    // ---- descriptors (Go2 command_direction config) ------------------------------------------------
    GfActionArgs aa{};
    aa.num_envs = N; aa.num_dofs = D; aa.mode = GF_ACTION_POSITION; aa.check_finite = 1;
    aa.actions_in = act_in; aa.scale = scale; aa.offset = def; aa.clip_lo = lo; aa.clip_hi = hi;
    aa.env_actions = act; aa.env_last_actions = last; aa.episode_length = ep; aa.targets = targets; aa.stats = stats;

    GfTerminationArgs ta{};
    ta.num_envs = N; ta.num_terms = 2; ta.entity = {pos, quat, lin, ang}; ta.episode_length = ep; ta.max_episode_length = maxlen;
    ta.terminated = term; ta.truncated = trunc; ta.stats = stats;
    ta.terms[0].op = GF_T_TIMEOUT; ta.terms[0].flags = GF_TERM_FLAG_TIME_OUT;
    ta.terms[1].op = GF_T_BAD_ORIENTATION; ta.terms[1].p[0] = std::sin(10.0f * 3.14159265f / 180.f); ta.terms[1].p[1] = 10.0f * 3.14159265f / 180.f;

    GfRewardArgs ra{};
    ra.num_envs = N; ra.num_dofs = D; ra.num_terms = T; ra.mode = GF_REWARD_MODE_STEP; ra.dt = 0.02f; ra.logging_enabled = 1;
    ra.entity = {pos, quat, lin, ang}; ra.dof_pos = dof; ra.default_dof_pos = def; ra.actions = act; ra.last_actions = last; ra.terminated = term;
    ra.command[0].command = cmd; ra.command[0].width = 3;
    ra.reward = reward; ra.episode_sums = sums; ra.episode_seconds = secs;
    int ops[6] = {GF_R_BASE_HEIGHT, GF_R_CMD_TRACK_LIN_VEL, GF_R_CMD_TRACK_ANG_VEL, GF_R_LIN_VEL_Z_L2, GF_R_ACTION_RATE_L2, GF_R_DOF_SIMILAR_TO_DEFAULT};
    float ws[6] = {-50.f * 0.02f, 1.f * 0.02f, 0.5f * 0.02f, -1.f * 0.02f, -0.005f * 0.02f, -0.1f * 0.02f};
    for (int k = 0; k < 6; ++k) { ra.terms[k].op = ops[k]; ra.terms[k].w = ws[k]; ra.terms[k].row = k; }
    ra.terms[0].p[0] = 0.3f; ra.terms[1].p[0] = 0.25f; ra.terms[2].p[0] = 0.25f; ra.terms[2].i[1] = 2;

    GfCommandArgs ca{};
    ca.num_envs = N; ca.num_ranges = 3; ca.mode = GF_CMD_STEP; ca.resample_steps = 250; ca.episode_length = ep; ca.seed = 1; ca.stream = 1;
    for (int i = 0; i < 3; ++i) { ca.lo[i] = -1.f; ca.hi[i] = 1.f; }
    ca.command = cmd; ca.stats = stats;
    GfCommandArgs cr = ca;
    cr.mode = GF_CMD_MASKED; cr.mask = term; cr.mask2 = trunc; cr.stats = nullptr;

    GfResetArgs rs{};
    rs.num_envs = N; rs.num_dofs = D; rs.num_reward_terms = T; rs.mask = term; rs.mask2 = trunc;
    rs.env_actions = act; rs.env_last_actions = last; rs.episode_length = ep; rs.max_episode_length = maxlen; rs.base_max_episode_length = 1000;
    rs.max_random_scaling = 100.f; rs.episode_sums = sums; rs.episode_seconds = secs; rs.reward_log_mask = 63; rs.reward_logging = 1;
    rs.scene_dof_pos = dof; rs.scene_dof_vel = dvel; rs.default_dof_pos = def; rs.scene_pos = pos; rs.scene_quat = quat; rs.quat_stash = stash;
    rs.scene_lin_vel = lin; rs.scene_ang_vel = ang; rs.reset_pos[2] = 0.4f; rs.reset_quat[0] = 1.f; rs.set_quat = 1; rs.zero_velocity = 1;
    rs.seed = 1; rs.stream = 2; rs.stats = stats;

    GfObservationArgs oa{};
    oa.num_envs = N; oa.num_dofs = D; oa.num_items = 7; oa.obs_width = O; oa.history_len = 1; oa.entity = {pos, quat, lin, ang};
    oa.dof_pos = dof; oa.dof_vel = dvel; oa.targets = targets; oa.env_actions = act; oa.command[0].command = cmd; oa.command[0].width = 3;
    oa.seed = 1; oa.stream = 3; oa.obs = obs; oa.stale_quat = stash; oa.stale_mask = term; oa.stale_mask2 = trunc;
    int oops[7] = {GF_O_COMMAND, GF_O_ANG_VEL_BODY, GF_O_LIN_VEL_BODY, GF_O_PROJ_GRAVITY, GF_O_DOF_POS, GF_O_DOF_VEL, GF_O_ACTIONS};
    int ow[7] = {3, 3, 3, 3, 12, 12, 12};
    for (int k = 0; k < 7; ++k) { oa.items[k].op = oops[k]; oa.items[k].width = ow[k]; oa.items[k].scale = 1.f; }
    oa.items[5].scale = 0.05f;

    GfSynthSceneArgs sa{};
    sa.num_envs = N; sa.num_dofs = D; sa.dt = 0.02f; sa.joint_rate = 10.f; sa.ang_noise = 0.05f; sa.lin_noise = 0.05f; sa.height_target = 0.4f;
    sa.targets = targets; sa.pos = pos; sa.quat = quat; sa.lin_vel = lin; sa.ang_vel = ang; sa.dof_pos = dof; sa.dof_vel = dvel; sa.seed = 9;

    auto chk = [](int rc, const char* what) { if (rc) { fprintf(stderr, "%s failed: %d %s\n", what, rc, gf_error_string(rc)); exit(2); } };
    chk(gf_termination_step(&ta, 0), "termination");  // fill masks once

    const double Nd = N;
    time_loop("empty<<<N/64,64>>>", iters, 0, [&] { empty_kernel<<<(N + 63) / 64, 64>>>(N); });
    // streaming reference: read 136 B/env + write 136 B/env through dedicated buffers sized for it
    const size_t copy4 = (size_t)N * 136 / 16;
    float4* cp_src = dalloc<float4>(copy4);
    float4* cp_dst = dalloc<float4>(copy4);
    time_loop("copy 272B/env (float4)", iters, 272.0 * Nd, [&] { copy_kernel<<<(unsigned)((copy4 + 255) / 256), 256>>>(cp_src, cp_dst, copy4); });
    time_loop("gf_action_step", iters, 248.0 * Nd, [&] { chk(gf_action_step(&aa, 0), "action"); });
    {   // as the recorded step runs it: zero the next statistics slot, fold the previous one into the vector ring
        GfStepStats* ring = dalloc<GfStepStats>(3 * GF_STATS_SHARDS);
        double* vec = dalloc<double>(64);
        double* last = dalloc<double>(64);
        GfActionArgs ar = aa;
        ar.stats = ring; ar.stats_zero = ring + GF_STATS_SHARDS; ar.stats_fold_src = ring + 2 * GF_STATS_SHARDS; ar.stats_fold_dst = vec; ar.stats_last_reset = last;
        time_loop("gf_action_step (+ring)", iters, 248.0 * Nd, [&] { chk(gf_action_step(&ar, 0), "action"); });
    }
    time_loop("gf_synth_scene_step", iters, 0, [&] { sa.tick++; chk(gf_synth_scene_step(&sa, 0), "scene"); });
    time_loop("gf_termination_step", iters, 26.0 * Nd, [&] { chk(gf_termination_step(&ta, 0), "termination"); });
    time_loop("gf_reward_step", iters, 268.0 * Nd, [&] { chk(gf_reward_step(&ra, 0), "reward"); });
    time_loop("gf_command_step", iters, 4.0 * Nd, [&] { chk(gf_command_step(&ca, 0), "command"); });
    time_loop("gf_masked_reset", iters, 2.0 * Nd, [&] { chk(gf_masked_reset(&rs, 0), "reset"); });
    time_loop("gf_command_step(masked)", iters, 2.0 * Nd, [&] { chk(gf_command_step(&cr, 0), "command"); });
    time_loop("gf_observe", iters, 388.0 * Nd, [&] { chk(gf_observe(&oa, 0), "observe"); });
    GfOp ops_[8] = {{GF_PHASE_ACTION, 0, &aa}, {GF_PHASE_SCENE, 0, &sa}, {GF_PHASE_TERMINATION, 0, &ta}, {GF_PHASE_REWARD, 0, &ra},
                    {GF_PHASE_COMMAND, 0, &ca}, {GF_PHASE_RESET, 0, &rs}, {GF_PHASE_COMMAND, 0, &cr}, {GF_PHASE_OBSERVE, 0, &oa}};
    GfPostRefs pr{};
    pr.termination = &ta; pr.reward = &ra; pr.reset = &rs; pr.num_command = 1; pr.num_observe = 1;
    pr.command_step[0] = &ca; pr.command_reset[0] = &cr; pr.observe[0] = &oa;
    cr.seed = rs.seed; ca.seed = rs.seed; oa.seed = rs.seed;
    {
        const int rc = gf_post_physics_check(&pr);
        printf("gf_post_physics_check: %d (%s)\n", rc, gf_error_string(rc));
        if (rc == 0) {
            gf_set_option(GF_OPT_POST_VARIANT, 0);
            time_loop("gf_post_physics_step (1 wave)", iters, 566.0 * Nd, [&] { chk(gf_post_physics_step(&pr, 0), "post"); });
            gf_set_option(GF_OPT_POST_VARIANT, 1);
            time_loop("gf_post_physics_step (4 waves)", iters, 566.0 * Nd, [&] { chk(gf_post_physics_step(&pr, 0), "post"); });
            gf_set_option(GF_OPT_POST_VARIANT, 2);
            char what[2048];
            gf_post_physics_describe(&pr, what, sizeof(what));
            printf("%s\n", what);
            time_loop("gf_post_physics_step (static)", iters, 566.0 * Nd, [&] { chk(gf_post_physics_step(&pr, 0), "post"); });
        }
    }
    int failed = -1;
    time_loop("full step (8 ops)", iters, 806.0 * Nd, [&] { sa.tick++; chk(gf_run_ops(ops_, 8, 0, &failed), "run_ops"); });
    return 0;
}
