// gf_post.hip — fused post-physics step: termination → reward → command.step → reset of done envs →
// command.reset → observations (managed_env.py:303-326) as ONE launch.
//
// Every phase after scene.step() is per-env work over the same state, so one lane carries its env through
// all of them with that state in registers:
//   * pos / quat / lin / ang / the [N,D] rows / commands / episode_length are loaded ONCE, up front, all
//     loads in flight together (straight-line load stream, zero-pad redirection, pinned kernarg pointers —
//     see gf_reward.hip);
//   * termination masks, the reward fold and the per-term episode sums (LDS-DMA prefetched) are evaluated by
//     the shared term bodies of gf_terms.h — the same code the per-phase kernels run;
//   * the reset of done envs is applied to the registers as well as to memory, so the observation that
//     follows needs no reload: it reads post-reset dof_pos / dof_vel / velocities / commands from registers,
//     and — reproducing the reference's stale EntityManager cache (entity_manager.py:163-167,189-195) — the
//     PRE-reset quaternion;
//   * the reward manager's reset (sum/seconds → log, sum ← 0) is folded into the sum update itself, which
//     removes a whole read-modify-write pass over the [T,N] sums;
//   * each wave assembles its [64, O] observation tile in LDS and streams it out with coalesced stores.
// Semantics are, by construction, those of calling the phase entry points in sequence (the oracle twin does
// exactly that); tests compare the two paths bit for bit.
// Algorithmic traffic, Go2 command config: R 13·4 + 5 rows·48 + cmd 12 + ep/max 8 + secs 4 + sums 24 = 340,
// W masks 2 + reward 4 + sums 24 + secs 4 + obs 192 = 226  →  566 B/env.
#include <dlfcn.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "gf_post_args.h"
#include "gf_post_ws.h"
#include "gf_post_programs.h"

namespace gf {

template <int DV>
__global__ __launch_bounds__(kEnvBlock) void post_kernel(const GfPostArgs karg) {
    // The 2.7 KB descriptor is staged into LDS once, with vector loads (all 42 lines in flight together), and every
    // later field read is a ds_read.  Reading it in place would cost one dependent, uncached scalar load per term /
    // item / field group: the kernarg block is rewritten by the host for every launch and lives in memory the scalar
    // cache does not keep, and in-kernel stamps showed ≈ 0.35 µs per table row — more than the arithmetic of the row.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kArgVec = (int)(sizeof(GfPostArgs) / 16);
    {
        const auto* src = (const __attribute__((address_space(4))) f32x4*)__builtin_amdgcn_kernarg_segment_ptr();
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        for (int i = threadIdx.x; i < kArgVec; i += kEnvBlock) dst[i] = src[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    const GfPostArgs& a = *reinterpret_cast<const GfPostArgs*>(lds);
    GF_STAMP(1);
    float* lds_sums = lds + kArgVec * 4;                     // [num_rew][64]
    float* lds_aux = lds_sums + kPostMaxReward * kEnvBlock;  // [kPostAuxRows][64] per-lane scratch for rolled Philox loops (dof reset noise)
    float* tile = lds_aux + kPostAuxRows * kEnvBlock;                  // [64][O+1]

    const int lane = threadIdx.x;
    const int64_t N = uni(a.num_envs);
    const int64_t n0 = (int64_t)blockIdx.x * kEnvBlock;
    const int64_t n_raw = n0 + lane;
    const bool live = n_raw < N;
    const int64_t n = live ? n_raw : N - 1;
    const uint32_t e = (uint32_t)n;
    const uint32_t genv = e + uni(a.env_offset);
    const int D = uni(a.num_dofs);
    const uint32_t needs = uni(a.needs);
    const int n_term = uni(a.num_term), n_rew = uni(a.num_rew), n_cmd = uni(a.n_cmd), n_obs = uni(a.n_obs);
    const uint64_t seed = uni(a.seed);
    constexpr int R = DV;

    // ---- 0. episode-sum columns: global -> LDS without VGPRs or waits ------------------------------------
    const bool logging = uni(a.logging) != 0;
    float* const k_sums = uni(a.episode_sums);
    if (logging)
        for (int k = 0; k < n_rew; ++k)
            __builtin_amdgcn_global_load_lds(k_sums + (int64_t)uni(a.rterms[k].row) * N + n, lds_sums + k * kEnvBlock, 4, 0, 0);

    // ---- 1. all per-env inputs, one straight-line burst ------------------------------------------------------
    const float *k_pos = uni(a.pos), *k_quat = uni(a.quat), *k_lin = uni(a.lin_vel), *k_ang = uni(a.ang_vel), *k_dof = uni(a.dof_pos),
                *k_dvel = uni(a.dof_vel);
    const float *k_tgt = uni(a.targets), *k_act = uni(a.env_actions), *k_last = uni(a.env_last_actions), *k_def = uni(a.default_dof_pos),
                *k_secs = uni(a.episode_seconds);
    const int32_t *k_ep = uni(a.episode_length), *k_max = uni(a.max_episode_length);

    const float4 q = ldg4(gsel((needs & PN_QUAT) != 0, k_quat, 4u * e));
    const GF_GLOBAL float* pp = gsel((needs & PN_POS) != 0, k_pos, 3u * e);
    const GF_GLOBAL float* lp = gsel((needs & PN_LIN) != 0, k_lin, 3u * e);
    const GF_GLOBAL float* ap = gsel((needs & PN_ANG) != 0, k_ang, 3u * e);
    V3 pos{pp[0], pp[1], pp[2]}, lin{lp[0], lp[1], lp[2]}, ang{ap[0], ap[1], ap[2]};
    const int ep_len = *gsel((needs & PN_EPLEN) != 0, k_ep, e);
    const int max_len = *gsel((needs & PN_MAXLEN) != 0, k_max, e);
    const float secs_in = *gsel(n_rew >= 0 && k_secs != nullptr, k_secs, e);

    const uint32_t ro = e * (uint32_t)D;
    float4 r_pos[R], r_vel[R], r_tgt[R], r_act[R], r_last[R], r_def[R];
    {
        const GF_GLOBAL float* p0 = gsel((needs & PN_DOFPOS) != 0, k_dof, ro);
        const GF_GLOBAL float* p1 = gsel((needs & PN_DOFVEL) != 0, k_dvel, ro);
        const GF_GLOBAL float* p2 = gsel((needs & PN_TARGETS) != 0, k_tgt, ro);
        const GF_GLOBAL float* p3 = gsel((needs & PN_ACTIONS) != 0, k_act, ro);
        const GF_GLOBAL float* p4 = gsel((needs & PN_LAST) != 0, k_last, ro);
        const GF_GLOBAL float* p5 = gsel(k_def != nullptr, k_def, 0u);
#pragma unroll
        for (int c = 0; c < DV; ++c) {
            r_pos[c] = ldg4(p0 + 4 * c); r_vel[c] = ldg4(p1 + 4 * c); r_tgt[c] = ldg4(p2 + 4 * c);
            r_act[c] = ldg4(p3 + 4 * c); r_last[c] = ldg4(p4 + 4 * c); r_def[c] = ldg4(p5 + 4 * c);
        }
    }
    // command rows of the stepped managers (≤ 2 managers × ≤ 4 ranges)
    float cmd[GF_POST_MAX_CMD][kPostMaxRanges];
#pragma unroll
    for (int c = 0; c < GF_POST_MAX_CMD; ++c) {
        const bool on = c < n_cmd;
        const uint32_t w = on ? (uint32_t)uni(a.cmds[c].width) : 0u;
        const GF_GLOBAL float* cp = gsel(on, on ? uni(a.cmds[c].command) : nullptr, e * w);
#pragma unroll
        for (int j = 0; j < kPostMaxRanges; ++j) cmd[c][j] = cp[(uint32_t)j < w ? j : 0];
    }

    GF_STAMP(2);
    // ---- 2. derived per-env quantities (pre-reset) ----------------------------------------------------------------
    const V3 blin = rot_inv(q, lin), bang = rot_inv(q, ang), grav = rot_inv(q, V3{0.f, 0.f, -1.f});
    float dof_dev = 0.f, act_rate = 0.f;
#pragma unroll
    for (int c = 0; c < DV; ++c) {
        dof_dev += fabsf(r_pos[c].x - r_def[c].x);
        dof_dev += fabsf(r_pos[c].y - r_def[c].y);
        dof_dev += fabsf(r_pos[c].z - r_def[c].z);
        dof_dev += fabsf(r_pos[c].w - r_def[c].w);
    }
#pragma unroll
    for (int c = 0; c < DV; ++c) {
        float d;
        d = r_last[c].x - r_act[c].x; act_rate += d * d;
        d = r_last[c].y - r_act[c].y; act_rate += d * d;
        d = r_last[c].z - r_act[c].z; act_rate += d * d;
        d = r_last[c].w - r_act[c].w; act_rate += d * d;
    }
    GfStepStats* const k_stats = uni(a.stats);
    GfStepStats* shard = k_stats ? stats_shard(k_stats) : nullptr;

    GF_STAMP(3);
    // ---- 3. termination (termination_manager.py:151-190) ---------------------------------------------------------------
    TermRegs tr;
    const int has_maxlen = uni(a.has_maxlen);
    tr.ep_len = ep_len; tr.max_len = max_len; tr.has_maxlen = has_maxlen != 0; tr.pos = pos; tr.m = n;
    tr.tilt_sin = clamp_max(norm2(grav.x, grav.y), 0.99f);
    int term = 0, trunc = 0;
    for (int k = 0; k < n_term; ++k) {
        const GfTerm t = a.tterms[k];
        int v = eval_termination_term(t, a, tr, (uint32_t)has_maxlen);
        v = live ? v : 0;
        if (t.flags & GF_TERM_FLAG_TIME_OUT) trunc |= v; else term |= v;
        if (shard) {
            const unsigned long long hit = __ballot(v);
            if (hit && lane == 0) atomicAdd(&shard->term_fired[k], popc64(hit));
        }
    }
    if (live) {
        G(uni(a.terminated))[n_raw] = (uint8_t)term;
        G(uni(a.truncated))[n_raw] = (uint8_t)trunc;
    }
    const bool done = live && (term | trunc);
    const unsigned long long done_mask = __ballot(done);
    if (shard && done_mask && lane == 0) atomicAdd(&shard->reset_count, popc64(done_mask));

    GF_STAMP(4);
    // ---- 4. reward (reward_manager.py:166-195) with the manager's reset (:197-222) folded into the sum update -----------
    float* const k_reward = uni(a.reward);
    if (n_rew >= 0 && k_reward) {
        RewardRegs rr;
        rr.pos = pos; rr.blin = blin; rr.bang = bang; rr.grav = grav; rr.dof_dev = dof_dev; rr.act_rate = act_rate; rr.terminated = term;
        rr.n = n; rr.live = live;
        const int c0 = uni(a.cmd_of_view[0]);
        if (c0 >= 0) {
            rr.cmd0[0] = c0 == 0 ? cmd[0][0] : cmd[1][0];
            rr.cmd0[1] = c0 == 0 ? cmd[0][1] : cmd[1][1];
            rr.cmd0[2] = c0 == 0 ? cmd[0][2] : cmd[1][2];
        } else {
            const float* v0 = uni(a.command[0].command);
            const bool nv = v0 != nullptr;
            const uint32_t w = nv ? (uint32_t)uni(a.command[0].width) : 0u;
            const uint32_t st = nv ? (uint32_t)(uni(a.command[0].stride) ? uni(a.command[0].stride) : uni(a.command[0].width)) : 0u;
            const GF_GLOBAL float* cp = gsel(nv, v0, e * st);
            rr.cmd0[0] = cp[0]; rr.cmd0[1] = cp[w > 1 ? 1 : 0]; rr.cmd0[2] = cp[w > 2 ? 2 : 0];
        }
        const float dt = uni(a.dt);
        const uint32_t log_mask = uni(a.reward_log_mask);
        const float secs_new = secs_in + dt;
        if (logging) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        float buf = 0.f;
        const bool log_reset = logging && done_mask != 0;
        for (int k = 0; k < n_rew; ++k) {
            const GfTerm t = a.rterms[k];
            float v = eval_reward_term(t, a, rr);
            v = v * t.w;
            buf += v;
            if (logging) {
                float s = lds_sums[k * kEnvBlock + lane] + v;
                if (log_reset) {
                    // RewardManager.reset: value /= seconds; mean → log; value ← 0 (zero-weight rows are not logged)
                    const float per_sec = done ? s / secs_new : 0.f;
                    if (shard && (log_mask & (1u << t.row))) {
                        if (popc64(done_mask) > 4) {
                            const double w = wave_sum((double)per_sec);
                            if (lane == 0) unsafeAtomicAdd(&shard->reward_episode_sum[t.row], w);
                        } else if (done) {
                            unsafeAtomicAdd(&shard->reward_episode_sum[t.row], (double)per_sec);
                        }
                    }
                    if (done) s = 0.f;
                }
                if (live) G(k_sums)[(int64_t)t.row * N + n_raw] = s;
            }
        }
        if (live) {
            G(k_reward)[n_raw] = buf;
            G(const_cast<float*>(k_secs))[n_raw] = done ? 1e-10f : secs_new;
        }
        if (done && logging)
            for (int row = 0; row < a.reward_rows; ++row)
                if (a.uncovered_rows & (1u << row)) G(k_sums)[(int64_t)row * N + n_raw] = 0.f;
    }

    GF_STAMP(5);
    // ---- 5. command.step: resample where episode_length % resample_steps == 0 (command_manager.py:152-162) --------------
    bool cmd_dirty[GF_POST_MAX_CMD] = {false, false};
#pragma unroll
    for (int c = 0; c < GF_POST_MAX_CMD; ++c) {
        if (c < n_cmd) {
            const PostCmd cm = a.cmds[c];
            const bool go = live && (ep_len % uni(cm.resample_steps)) == 0;
            if (shard) {
                const unsigned long long m = __ballot(go);
                if (m && lane == 0) atomicAdd(&shard->resample_count, popc64(m));
            }
            if (go) {
                const float4 u4 = draw_unit4(seed, cm.stream_step, genv, 0u);
                const float nv[kPostMaxRanges] = {uniform_range(u4.x, cm.lo[0], cm.hi[0]), uniform_range(u4.y, cm.lo[1], cm.hi[1]),
                                                  uniform_range(u4.z, cm.lo[2], cm.hi[2]), uniform_range(u4.w, cm.lo[3], cm.hi[3])};
#pragma unroll
                for (int j = 0; j < kPostMaxRanges; ++j)
                    if (j < cm.width) cmd[c][j] = nv[j];
                cmd_dirty[c] = true;
            }
        }
    }

    GF_STAMP(6);
    // ---- 6. reset of done envs (managed_env.py:336-366), applied to memory AND to the registers the observation reads ----
    float4 o_act[R];  // raw actions as the observation sees them
#pragma unroll
    for (int c = 0; c < DV; ++c) o_act[c] = r_act[c];
    if (done) {
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((a.reset_env & 1) && a.env_actions) {
            GF_GLOBAL f32x4* ra = reinterpret_cast<GF_GLOBAL f32x4*>(G(const_cast<float*>(k_act)) + n * D);
            GF_GLOBAL f32x4* rl = reinterpret_cast<GF_GLOBAL f32x4*>(G(const_cast<float*>(k_last)) + n * D);
#pragma unroll
            for (int c = 0; c < DV; ++c) { ra[c] = f32x4{0.f, 0.f, 0.f, 0.f}; rl[c] = f32x4{0.f, 0.f, 0.f, 0.f}; o_act[c] = z4; }
        }
        if (a.reset_env & 2) G(const_cast<int32_t*>(k_ep))[n] = 0;
        if (a.max_episode_length && a.max_random_scaling > 0.0f) {
            const float u = draw_unit4(seed, a.stream_reset, genv, 0u).x;
            const float rnd = uniform_range(u, -1.0f, 1.0f) * a.max_random_scaling;
            G(const_cast<int32_t*>(k_max))[n] = (int32_t)rintf((float)a.base_max_episode_length + rnd);
        }
        for (int m = 0; m < a.n_air; ++m) {
            const int L = a.air_links[m];
            for (int s = 0; s < 4; ++s) {
                GF_GLOBAL float* p = G(a.air_state[m][s]);
                if (p)
                    for (int l = 0; l < L; ++l) p[n * L + l] = 0.0f;
            }
        }
        if (a.reset_dofs) {
            GF_GLOBAL f32x4* dp = reinterpret_cast<GF_GLOBAL f32x4*>(G(const_cast<float*>(k_dof)) + n * D);
            GF_GLOBAL f32x4* dv = reinterpret_cast<GF_GLOBAL f32x4*>(G(const_cast<float*>(k_dvel)) + n * D);
            const float dof_noise = a.dof_noise_scale;
            if (dof_noise != 0.0f) {
                // one Philox block yields the four draws of columns 4+4c .. 4+4c+3 (they share counter (4+4c)>>2)
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
                float4 p = r_def[c];
                if (dof_noise != 0.0f) {
                    const float* sc = lds_aux + (4 * c) * kEnvBlock + lane;
                    p.x = p.x + sc[0 * kEnvBlock];
                    p.y = p.y + sc[1 * kEnvBlock];
                    p.z = p.z + sc[2 * kEnvBlock];
                    p.w = p.w + sc[3 * kEnvBlock];
                }
                r_pos[c] = p;
                dp[c] = f32x4{p.x, p.y, p.z, p.w};
                if (k_dvel) { dv[c] = f32x4{0.f, 0.f, 0.f, 0.f}; r_vel[c] = z4; }
            }
        }
        if (a.scene_reset) {
            GF_GLOBAL float* wp = G(const_cast<float*>(k_pos)) + 3 * n;
            float np[3] = {a.reset_pos[0], a.reset_pos[1], a.reset_pos[2]};
            float4 nq = make_float4(a.reset_quat[0], a.reset_quat[1], a.reset_quat[2], a.reset_quat[3]);
            bool set_quat = a.set_quat != 0;
            if (a.spawn_mode) {  // mdp.reset.randomize_terrain_position
                float u[5];
                spawn_draws(nullptr, n, seed, a.stream_reset, genv, a.spawn_rot_mask, u);
                spawn_pose(a, u, np, &nq);
                set_quat = a.spawn_set_quat != 0;
            }
            wp[0] = np[0]; wp[1] = np[1]; wp[2] = np[2];
            if (set_quat) {
                if (a.quat_stash) reinterpret_cast<GF_GLOBAL f32x4*>(G(a.quat_stash))[n] = f32x4{q.x, q.y, q.z, q.w};
                reinterpret_cast<GF_GLOBAL f32x4*>(G(const_cast<float*>(k_quat)))[n] = f32x4{nq.x, nq.y, nq.z, nq.w};
            }
            if (a.zero_velocity) {
                GF_GLOBAL float* wl = G(const_cast<float*>(k_lin)) + 3 * n;
                GF_GLOBAL float* wa = G(const_cast<float*>(k_ang)) + 3 * n;
                wl[0] = 0.f; wl[1] = 0.f; wl[2] = 0.f;
                wa[0] = 0.f; wa[1] = 0.f; wa[2] = 0.f;
                lin = V3{0.f, 0.f, 0.f};
                ang = V3{0.f, 0.f, 0.f};
                if (k_dvel) {
                    GF_GLOBAL f32x4* dv = reinterpret_cast<GF_GLOBAL f32x4*>(G(const_cast<float*>(k_dvel)) + n * D);
#pragma unroll
                    for (int c = 0; c < DV; ++c) { dv[c] = f32x4{0.f, 0.f, 0.f, 0.f}; r_vel[c] = z4; }
                }
            }
        }
    }

    // ---- 7. command.reset for done envs (command_manager.py:164-170) + write back changed commands ------------------------
#pragma unroll
    for (int c = 0; c < GF_POST_MAX_CMD; ++c) {
        if (c < n_cmd) {
            const PostCmd cm = a.cmds[c];
            if (done) {
                const float4 u4 = draw_unit4(seed, cm.stream_reset, genv, 0u);
                const float nv[kPostMaxRanges] = {uniform_range(u4.x, cm.lo[0], cm.hi[0]), uniform_range(u4.y, cm.lo[1], cm.hi[1]),
                                                  uniform_range(u4.z, cm.lo[2], cm.hi[2]), uniform_range(u4.w, cm.lo[3], cm.hi[3])};
#pragma unroll
                for (int j = 0; j < kPostMaxRanges; ++j)
                    if (j < cm.width) cmd[c][j] = nv[j];
                cmd_dirty[c] = true;
            }
            if (cmd_dirty[c] && live) {
                GF_GLOBAL float* crow = G(cm.command) + n * cm.width;
#pragma unroll
                for (int j = 0; j < kPostMaxRanges; ++j)
                    if (j < cm.width) crow[j] = cmd[c][j];
            }
        }
    }

    GF_STAMP(7);
    // ---- 8. observations (observation_manager.py:218-256): post-reset state, pre-reset quaternion ------------------------
    // a done env's velocities were zeroed above and rot_inv(q, 0) is exactly +0: no second rotation needed
    const bool zeroed = done && a.scene_reset && a.zero_velocity;
    const V3 o_lin = zeroed ? V3{0.f, 0.f, 0.f} : blin, o_ang = zeroed ? V3{0.f, 0.f, 0.f} : bang;
    for (int m = 0; m < n_obs; ++m) {
        const PostObs& ob = a.obs[m];
        const int O = uni(ob.width), S = O + 1, H = uni(ob.history), n_items = uni(ob.num_items);
        float* const ob_out = uni(ob.obs);
        const float* const ob_prev = uni(ob.prev);
        const uint64_t ob_stream = uni(ob.stream);
        float* row = tile + lane * S;
        int col = 0;
        for (int i = 0; i < n_items; ++i) {
            const GfObsItem it = ob.items[i];  // by value: the tile stores below must not force reloads of the item
            const float it_scale = uni(it.scale), it_noise = uni(it.noise);
            const ObsFin f{it_scale, it_scale != 1.0f};
            switch (uni(it.op)) {
                case GF_O_COMMAND: {
                    const int owner = a.cmd_of_view[it.i0];
                    if (owner >= 0) {
#pragma unroll
                        for (int j = 0; j < kPostMaxRanges; ++j)
                            if (j < it.width) row[col + j] = obs_finish(f, owner == 0 ? cmd[0][j] : cmd[1][j], col + j);
                    } else {
                        const GfCommandView cv = a.command[it.i0];
                        for (int j = 0; j < it.width; ++j) row[col + j] = obs_finish(f, G(cv.command)[n * cmd_stride(cv) + j], col + j);
                    }
                } break;
                case GF_O_ANG_VEL_BODY:
                case GF_O_LIN_VEL_BODY:
                case GF_O_PROJ_GRAVITY: {
                    const V3 v = it.op == GF_O_ANG_VEL_BODY ? o_ang : (it.op == GF_O_LIN_VEL_BODY ? o_lin : grav);
                    row[col + 0] = obs_finish(f, v.x, col + 0);
                    row[col + 1] = obs_finish(f, v.y, col + 1);
                    row[col + 2] = obs_finish(f, v.z, col + 2);
                } break;
                case GF_O_DOF_POS: put_row<DV>(f, r_pos, row, col, 4 * DV); break;
                case GF_O_DOF_VEL: put_row<DV>(f, r_vel, row, col, 4 * DV); break;
                case GF_O_ACTIONS: put_row<DV>(f, r_tgt, row, col, 4 * DV); break;
                case GF_O_RAW_ACTIONS: put_row<DV>(f, o_act, row, col, 4 * DV); break;
                case GF_O_DOF_FORCE: {
                    const GF_GLOBAL float* r = G(a.dof_force) + n * D;
                    for (int j = 0; j < it.width; ++j) row[col + j] = obs_finish(f, r[j], col + j);
                } break;
                case GF_O_CONTACT_FORCE_NORM: {
                    const GfContactView cv = a.contact[it.i0];
                    const GF_GLOBAL float* r = G(cv.contacts) + n * cv.num_links * 3;
                    for (int l = 0; l < it.width; ++l) row[col + l] = obs_finish(f, norm3(r[3 * l], r[3 * l + 1], r[3 * l + 2]), col + l);
                } break;
                default: break;
            }
            const int it_w = uni(it.width);
            if (it_noise != 0.0f) {
                // += uniform_(-1,1)*noise on the scaled value (observation_manager.py:247-250); kept as a rolled loop so the
                // kernel contains ONE inlined Philox here instead of one per observation element
                // (column c draws word c & 3 of Philox block c >> 2: one block per four columns the item covers)
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
            col += it_w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        GF_STAMP(8);

        const int rows = (int)((N - n0) < kEnvBlock ? (N - n0) : kEnvBlock);
        const int64_t OH = (int64_t)O * H;
        GF_GLOBAL float* out = G(ob_out) + n0 * OH;
        if ((O & 3) == 0) {
            const int o4 = O >> 2;
            const int qstep = GF_WAVE / o4, rstep = GF_WAVE - qstep * o4;  // wave-uniform: one division per tile, not per element
            int rw = lane / o4, c4 = lane - rw * o4;
            for (int i = lane; i < rows * o4; i += GF_WAVE) {
                const float* r = tile + rw * S + c4 * 4;
                reinterpret_cast<GF_GLOBAL f32x4*>(out + rw * OH)[c4] = f32x4{r[0], r[1], r[2], r[3]};
                rw += qstep; c4 += rstep;
                if (c4 >= o4) { c4 -= o4; ++rw; }
            }
            if (H > 1) {
                const int h4 = (O * (H - 1)) >> 2;
                const GF_GLOBAL float* prev = G(ob_prev) + n0 * OH;
                for (int i = lane; i < rows * h4; i += GF_WAVE) {
                    const int rw = i / h4, j = i - rw * h4;
                    reinterpret_cast<GF_GLOBAL f32x4*>(out + rw * OH + O)[j] = reinterpret_cast<const GF_GLOBAL f32x4*>(prev + rw * OH)[j];
                }
            }
        } else {
            for (int i = lane; i < rows * O; i += GF_WAVE) {
                const int rw = i / O, cc = i - rw * O;
                out[rw * OH + cc] = tile[rw * S + cc];
            }
            if (H > 1) {
                const int hw = O * (H - 1);
                const GF_GLOBAL float* prev = G(ob_prev) + n0 * OH;
                for (int i = lane; i < rows * hw; i += GF_WAVE) {
                    const int rw = i / hw, j = i - rw * hw;
                    out[rw * OH + O + j] = prev[rw * OH + j];
                }
            }
        }
        // the tile is reused by the next observation manager
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
    GF_STAMP(9);
}

// ------------------------------------------------------------------------------------------------------------
// Host side: validate that the per-phase descriptors describe one fusable step and pack them.
// ------------------------------------------------------------------------------------------------------------
static int cmd_slot_of_reward_op(int op) {
    switch (op) {
        case GF_R_CMD_TRACK_LIN_VEL:
        case GF_R_CMD_TRACK_ANG_VEL:
        case GF_R_STAND_STILL: return 0;
        case GF_R_FEET_AIR_TIME: return 1;
        default: return -1;
    }
}
static bool reward_op_has_contact(int op) { return op == GF_R_HAS_CONTACT || op == GF_R_CONTACT_FORCE || op == GF_R_FEET_AIR_TIME || op == GF_R_FEET_SLIDE; }
static bool term_op_has_contact(int op) { return op == GF_T_HAS_CONTACT || op == GF_T_CONTACT_FORCE || op == GF_T_CONTACT_FORCE_GRACE; }

struct Packer {
    GfPostArgs a{};
    int n_contact = 0, n_view = 0;

    int contact_slot(const GfContactView& v) {
        for (int k = 0; k < n_contact; ++k)
            if (a.contact[k].contacts == v.contacts) {
                if (!a.contact[k].link_vel) a.contact[k].link_vel = v.link_vel;
                if (!a.contact[k].link_pos) a.contact[k].link_pos = v.link_pos;
                return k;
            }
        if (n_contact >= GF_MAX_CONTACT_VIEWS) return -1;
        a.contact[n_contact] = v;
        return n_contact++;
    }
    int view_slot(const GfCommandView& v) {
        for (int k = 0; k < n_view; ++k)
            if (a.command[k].command == v.command && a.command[k].width == v.width && a.command[k].stride == v.stride) return k;
        if (n_view >= GF_MAX_COMMAND_VIEWS) return -1;
        a.command[n_view] = v;
        a.cmd_of_view[n_view] = -1;
        if (a.n_gait && v.command == a.gait.state) {
            // a view of the gait manager's state rows (observation(): 14 columns of the 16-float row): travels through LDS
            if (v.stride != GF_GAIT_ROW || v.width > GF_GAIT_ROW) return -1;
            a.cmd_of_view[n_view] = kViewGait;
            return n_view++;
        }
        for (int c = 0; c < a.n_cmd; ++c)
            if (a.cmds[c].command == v.command) {
                // a fused command manager's own buffer is dense and travels through registers / LDS: a strided alias of it
                // (a column view of a resampled command) stays on the phase-by-phase path
                if (v.stride && v.stride != v.width) return -1;
                a.cmd_of_view[n_view] = c;
            }
        return n_view++;
    }
};

static bool same_entity(const GfEntityView& x, const GfEntityView& y) {
    auto ok = [](const float* p, const float* q) { return !p || !q || p == q; };
    return ok(x.pos, y.pos) && ok(x.quat, y.quat) && ok(x.lin_vel, y.lin_vel) && ok(x.ang_vel, y.ang_vel);
}
static void merge_entity(GfPostArgs& a, const GfEntityView& v) {
    if (v.pos) a.pos = const_cast<float*>(v.pos);
    if (v.quat) a.quat = const_cast<float*>(v.quat);
    if (v.lin_vel) a.lin_vel = const_cast<float*>(v.lin_vel);
    if (v.ang_vel) a.ang_vel = const_cast<float*>(v.ang_vel);
}

// GF_POST_WHY=1 in the environment names the rule that kept a step from fusing (development aid; checked once)
static bool why_enabled() {
    static const bool on = getenv("GF_POST_WHY") != nullptr;
    return on;
}
#define UNSUP(cond)                                                                                  \
    do {                                                                                             \
        if (cond) {                                                                                  \
            if (why_enabled()) fprintf(stderr, "gf_post_physics: not fusable (%s:%d): %s\n", __FILE__, __LINE__, #cond); \
            return GF_E_UNSUPPORTED;                                                                 \
        }                                                                                            \
    } while (0)

static int pack(const GfPostRefs* r, Packer& pk) {
    if (!r || !r->termination) return GF_E_NULL;
    // GF_POST_NO_RESET: termination … command / gait step, nothing behind them.  There is no reset to describe, so `reset` and the
    // masked command / gait descriptors may be absent; the checks below then run against a stand-in that says "no section of the
    // reset applies" (the seed / env offset every phase shares comes from the first stepped manager).
    const bool no_reset = (r->flags & GF_POST_NO_RESET) != 0;
    if (!no_reset && !r->reset) return GF_E_NULL;
    GfPostArgs& a = pk.a;
    const GfTerminationArgs& T = *r->termination;
    GfResetArgs none{};
    if (no_reset && !r->reset) {
        none.num_envs = T.num_envs; none.mask = T.terminated; none.mask2 = T.truncated;
        none.num_dofs = r->reward ? r->reward->num_dofs : 0;
        if (r->num_command > 0 && r->command_step[0]) { none.seed = r->command_step[0]->seed; none.env_offset = r->command_step[0]->env_offset; }
        else if (r->num_gait > 0 && r->gait_step[0]) { none.seed = r->gait_step[0]->seed; none.env_offset = r->gait_step[0]->env_offset; }
    }
    const GfResetArgs& RS = r->reset ? *r->reset : none;
    const GfRewardArgs* RW = r->reward;
    const int N = T.num_envs;
    UNSUP(no_reset && (r->num_observe != 0 || r->rollout || (r->flags & GF_POST_OBSERVE_ONLY)));
    a.no_reset = no_reset ? 1 : 0;
    // GF_POST_OBSERVE_ONLY: the step's phases up to the reset have run as launches of their own; `reset` is the descriptor that reset
    // ran with (its masks, its stale-quaternion stash, the seed every phase shares) and nothing in it is applied again
    const bool obs_only = (r->flags & GF_POST_OBSERVE_ONLY) != 0;
    UNSUP(obs_only && (r->reward || r->num_command || r->num_gait || r->rollout || r->num_observe < 1));
    a.obs_only = obs_only ? 1 : 0;
    UNSUP(N <= 0 || T.num_terms > kPostMaxTerm || T.term_out);
    UNSUP(r->num_command < 0 || r->num_command > GF_POST_MAX_CMD || r->num_observe < 0 || r->num_observe > GF_POST_MAX_OBS);
    UNSUP(RS.num_envs != N || RS.mask != T.terminated || RS.mask2 != T.truncated);
    UNSUP(RS.len_draws || RS.dof_draws);
    bool have_terrain = false;  // one terrain map per fused step: the reward's and the spawn's must be the same
    a.num_envs = N;
    a.terminated = T.terminated; a.truncated = T.truncated;
    a.stats = obs_only ? nullptr : (T.stats ? T.stats : RS.stats);   // (an observation-only launch counts nothing)
    UNSUP(!obs_only && RS.stats && T.stats && RS.stats != T.stats);
    a.episode_length = const_cast<int32_t*>(T.episode_length);
    a.max_episode_length = const_cast<int32_t*>(T.max_episode_length);
    a.has_maxlen = T.max_episode_length != nullptr;
    merge_entity(a, T.entity);
    uint32_t needs = 0;
    int D = RS.num_dofs;

    // commands first (views map onto them)
    a.n_cmd = r->num_command;
    for (int c = 0; c < a.n_cmd; ++c) {
        const GfCommandArgs* s = r->command_step[c];
        const GfCommandArgs* m = r->command_reset[c];
        UNSUP(!s || (!m && !no_reset) || s->mode != GF_CMD_STEP);
        UNSUP(s->num_envs != N || s->num_ranges > kPostMaxRanges || s->draws || s->resample_steps <= 0 || s->episode_length != T.episode_length);
        UNSUP(s->seed != RS.seed || s->env_offset != RS.env_offset);
        if (m) {
            UNSUP(m->mode != GF_CMD_MASKED || s->command != m->command || s->num_ranges != m->num_ranges || m->num_envs != N || m->draws);
            UNSUP(m->mask != T.terminated || m->mask2 != T.truncated || m->seed != RS.seed || m->env_offset != RS.env_offset);
        }
        UNSUP(s->stats && a.stats && s->stats != a.stats);
        PostCmd& pc = a.cmds[c];
        pc.command = s->command; pc.width = s->num_ranges; pc.resample_steps = s->resample_steps;
        pc.stream_step = s->stream; pc.stream_reset = m ? m->stream : 0;
        for (int j = 0; j < s->num_ranges; ++j) {
            UNSUP(m && (s->lo[j] != m->lo[j] || s->hi[j] != m->hi[j]));
            pc.lo[j] = s->lo[j]; pc.hi[j] = s->hi[j];
        }
        needs |= PN_EPLEN;
    }
    a.seed = RS.seed; a.env_offset = RS.env_offset; a.stream_reset = RS.stream;

    // the GaitCommandManager (examples/gait_trainer): stepped and reset on wave 0's registers
    UNSUP(r->num_gait < 0 || r->num_gait > GF_POST_MAX_GAIT);
    a.n_gait = r->num_gait;
    if (a.n_gait) {
        const GfGaitArgs* gs = r->gait_step[0];
        const GfGaitArgs* gm = r->gait_reset[0];
        UNSUP(!gs || (!gm && !no_reset) || gs->mode != GF_CMD_STEP || !gs->state || !gs->selected);
        UNSUP(gs->num_envs != N || gs->draws || gs->resample_steps <= 0 || gs->num_gaits < 1 || gs->num_gaits > GF_MAX_GAITS || gs->episode_length != T.episode_length);
        UNSUP(gs->seed != RS.seed || gs->env_offset != RS.env_offset);
        UNSUP(gs->stats && a.stats && gs->stats != a.stats);
        UNSUP((gs->wave_flags != nullptr) != (r->gait_flags_next[0] != nullptr) || (gs->wave_flags && gs->wave_flags == r->gait_flags_next[0]));
        UNSUP(reinterpret_cast<uintptr_t>(gs->state) & 15u);
        if (gm) {
            UNSUP(gm->mode != GF_CMD_MASKED || gs->state != gm->state || gs->selected != gm->selected || gm->num_envs != N || gm->draws);
            UNSUP(gm->mask != T.terminated || gm->mask2 != T.truncated || gm->seed != RS.seed || gm->env_offset != RS.env_offset);
            UNSUP(gs->wave_flags != gm->wave_flags);
            // both descriptors are filled from the same manager state (curriculum values are re-read per launch)
            UNSUP(gs->num_gaits != gm->num_gaits || gs->fixed_clearance_mask != gm->fixed_clearance_mask || gs->dt != gm->dt || gs->two_pi != gm->two_pi);
            UNSUP(memcmp(gs->cum_weight, gm->cum_weight, sizeof(gs->cum_weight)) != 0 || memcmp(gs->gait_offsets, gm->gait_offsets, sizeof(gs->gait_offsets)) != 0);
            UNSUP(gs->clearance_lo != gm->clearance_lo || gs->clearance_hi != gm->clearance_hi || gs->period_lo != gm->period_lo || gs->period_hi != gm->period_hi);
        }
        PostGait& pg = a.gait;
        pg.state = gs->state; pg.selected = gs->selected; pg.flags_in = gs->wave_flags; pg.flags_out = r->gait_flags_next[0];
        pg.stream_step = gs->stream; pg.stream_reset = gm ? gm->stream : 0;
        pg.resample_steps = gs->resample_steps; pg.num_gaits = gs->num_gaits; pg.fixed_clearance_mask = gs->fixed_clearance_mask;
        memcpy(pg.cum_weight, gs->cum_weight, sizeof(pg.cum_weight));
        memcpy(pg.gait_offsets, gs->gait_offsets, sizeof(pg.gait_offsets));
        pg.clearance_lo = gs->clearance_lo; pg.clearance_hi = gs->clearance_hi; pg.period_lo = gs->period_lo; pg.period_hi = gs->period_hi;
        pg.dt = gs->dt; pg.two_pi = gs->two_pi;
        needs |= PN_EPLEN;
    }

    // termination terms (not evaluated when the phase has already run as a launch of its own: the masks are inputs then)
    a.term_done = ((r->flags & GF_POST_TERMINATION_DONE) || obs_only) ? 1 : 0;
    a.num_term = a.term_done ? 0 : T.num_terms;
    for (int k = 0; k < a.num_term; ++k) {
        GfTerm t = T.terms[k];
        switch (t.op) {
            case GF_T_TIMEOUT: if (T.max_episode_length) needs |= PN_EPLEN | PN_MAXLEN; break;
            case GF_T_BAD_ORIENTATION: needs |= PN_QUAT | PN_EPLEN; break;
            case GF_T_BASE_HEIGHT_BELOW:
            case GF_T_OUT_OF_BOUNDS: needs |= PN_POS; break;
            case GF_T_CONTACT_FORCE_GRACE: needs |= PN_EPLEN;  // fallthrough
            case GF_T_HAS_CONTACT:
            case GF_T_CONTACT_FORCE: break;
            default: return GF_E_UNSUPPORTED;  // GF_T_EXTERNAL: a host-evaluated column needs GF_POST_TERMINATION_DONE (the table is then not evaluated here)
        }
        if (term_op_has_contact(t.op)) {
            UNSUP(t.i[0] < 0 || t.i[0] >= GF_MAX_CONTACT_VIEWS || !T.contact[t.i[0]].contacts);
            const int s = pk.contact_slot(T.contact[t.i[0]]);
            UNSUP(s < 0);
            t.i[0] = s;
        }
        a.tterms[k] = t;
    }

    // reward
    a.num_rew = -1;
    if (RW) {
        UNSUP(RW->num_envs != N || RW->mode != GF_REWARD_MODE_STEP || RW->num_terms > kPostMaxReward || !RW->reward || !RW->episode_seconds);
        UNSUP(!same_entity(T.entity, RW->entity));
        merge_entity(a, RW->entity);
        a.num_rew = RW->num_terms;
        a.reward = RW->reward; a.episode_sums = RW->episode_sums; a.episode_seconds = RW->episode_seconds;
        a.logging = RW->logging_enabled && RW->episode_sums;
        a.dt = RW->dt;
        for (int s = 0; s < 4; ++s) a.state[s] = RW->state[s];
        // view 0 first so the kernel's preloaded cmd0 refers to it
        if (RW->command[0].command) UNSUP(pk.view_slot(RW->command[0]) != 0);
        uint32_t covered = 0;
        for (int k = 0; k < RW->num_terms; ++k) {
            GfTerm t = RW->terms[k];
            UNSUP(t.row < 0 || t.row >= 24);
            covered |= 1u << t.row;
            switch (t.op) {
                case GF_R_IS_ALIVE:
                case GF_R_TERMINATED: UNSUP(RW->terminated != T.terminated); break;
                case GF_R_BASE_HEIGHT:
                    needs |= PN_POS;
                    if (t.flags & GF_RW_FLAG_TERRAIN) {
                        UNSUP(RW->terrain.height_field && (RW->terrain.rows < 1 || RW->terrain.cols < 1));
                        UNSUP(have_terrain && memcmp(&a.terrain, &RW->terrain, sizeof(GfTerrainView)) != 0);
                        a.terrain = RW->terrain;
                        have_terrain = true;
                    }
                    break;
                case GF_R_DOF_SIMILAR_TO_DEFAULT:
                case GF_R_STAND_STILL: needs |= PN_DOFPOS | PN_DOFDEV; break;
                case GF_R_LIN_VEL_Z_L2: needs |= PN_QUAT | PN_LIN; break;
                case GF_R_ANG_VEL_XY_L2: needs |= PN_QUAT | PN_ANG; break;
                case GF_R_FLAT_ORIENTATION_L2: needs |= PN_QUAT; break;
                case GF_R_BODY_ACCEL_EXP: needs |= PN_QUAT | PN_LIN | PN_ANG; UNSUP(t.i[0] < 0 || t.i[0] >= 4 || !RW->state[t.i[0]]); break;
                case GF_R_ACTION_RATE_L2: needs |= PN_ACTIONS | PN_LAST | PN_ACTRATE; break;
                case GF_R_CMD_TRACK_LIN_VEL: needs |= PN_QUAT | PN_LIN; break;
                case GF_R_CMD_TRACK_ANG_VEL: needs |= PN_QUAT | PN_ANG; break;
                case GF_R_HAS_CONTACT:
                case GF_R_CONTACT_FORCE:
                case GF_R_FEET_AIR_TIME:
                case GF_R_FEET_SLIDE: break;
                case GF_R_GAIT_PHASE:
                case GF_R_FOOT_HEIGHT: {
                    // read the feet's contact / velocity / position buffers and the PRE-step gait rows straight from memory
                    UNSUP(t.i[0] < 0 || t.i[0] >= GF_MAX_CONTACT_VIEWS || !RW->contact[t.i[0]].contacts || !RW->contact[t.i[0]].link_vel);
                    UNSUP(t.op == GF_R_FOOT_HEIGHT && !RW->contact[t.i[0]].link_pos);
                    UNSUP(t.i[1] < 0 || t.i[1] >= GF_MAX_COMMAND_VIEWS || !RW->command[t.i[1]].command || RW->command[t.i[1]].stride != GF_GAIT_ROW);
                    for (int f = 0; f < 4; ++f) UNSUP(((t.i[2] >> (8 * f)) & 0xff) >= RW->contact[t.i[0]].num_links);
                    const int cs2 = pk.contact_slot(RW->contact[t.i[0]]);
                    const int vs2 = pk.view_slot(RW->command[t.i[1]]);
                    UNSUP(cs2 < 0 || vs2 < 0);
                    t.i[0] = cs2; t.i[1] = vs2;
                    if (t.op == GF_R_GAIT_PHASE && RW->gait_wave_flags) {
                        UNSUP(!a.n_gait || RW->gait_wave_flags != a.gait.flags_in);   // the bytes this launch may read are the ones it does not write
                        a.gait_wave_flags = RW->gait_wave_flags;
                    }
                } break;
                case GF_R_EXTERNAL:   // a column the host evaluated after the termination phase: only behind GF_POST_TERMINATION_DONE
                    UNSUP(!a.term_done || t.i[0] < 0 || t.i[0] >= GF_MAX_EXT || !RW->ext[t.i[0]]);
                    a.ext[t.i[0]] = RW->ext[t.i[0]];
                    break;
                default: return GF_E_UNSUPPORTED;
            }
            if (t.op == GF_R_BASE_HEIGHT && (t.flags & GF_RW_FLAG_CMD)) {
                UNSUP(t.i[0] < 0 || t.i[0] >= GF_MAX_COMMAND_VIEWS || !RW->command[t.i[0]].command);
                const int s = pk.view_slot(RW->command[t.i[0]]);
                UNSUP(s < 0 || pk.a.cmd_of_view[s] >= 0);  // a resampled buffer as height target: keep the unfused path
                t.i[0] = s;
            }
            const int cs = cmd_slot_of_reward_op(t.op);
            if (cs >= 0 && t.i[cs] >= 0) {
                UNSUP(t.i[cs] >= GF_MAX_COMMAND_VIEWS || !RW->command[t.i[cs]].command);
                const int s = pk.view_slot(RW->command[t.i[cs]]);
                UNSUP(s < 0);
                // terms read view 0 from registers and any other view from memory — as it is BEFORE this step's resample: every
                // command / gait row the launch rewrites is stored behind the barrier the reward wave passes after its fold
                t.i[cs] = s;
            }
            if (reward_op_has_contact(t.op)) {
                UNSUP(t.i[0] < 0 || t.i[0] >= GF_MAX_CONTACT_VIEWS || !RW->contact[t.i[0]].contacts);
                const int s = pk.contact_slot(RW->contact[t.i[0]]);
                UNSUP(s < 0);
                t.i[0] = s;
            }
            a.rterms[k] = t;
        }
        if (needs & (PN_DOFPOS | PN_DOFDEV)) {
            UNSUP(!RW->dof_pos || !RW->default_dof_pos);
            a.dof_pos = const_cast<float*>(RW->dof_pos);
            a.default_dof_pos = RW->default_dof_pos;
            D = RW->num_dofs;
        }
        if (needs & PN_ACTIONS) {
            UNSUP(!RW->actions || !RW->last_actions);
            a.env_actions = const_cast<float*>(RW->actions);
            a.env_last_actions = const_cast<float*>(RW->last_actions);
            D = RW->num_dofs;
        }
        // reward-manager reset section must address the same buffers
        UNSUP(RS.episode_seconds && RS.episode_seconds != RW->episode_seconds);
        UNSUP(RS.episode_sums && RS.episode_sums != RW->episode_sums);
        UNSUP(!no_reset && !RS.episode_seconds);  // the fused kernel always resets the seconds of done envs
        UNSUP(!no_reset && (RS.reward_logging != 0) != (a.logging != 0));
        a.reward_rows = RS.num_reward_terms;
        a.reward_log_mask = RS.reward_log_mask;
        a.uncovered_rows = 0;
        for (int row = 0; row < RS.num_reward_terms; ++row)
            if (!(covered & (1u << row))) a.uncovered_rows |= 1u << row;
    } else if (!obs_only) {
        UNSUP(RS.episode_seconds || RS.episode_sums);
    }

    // reset sections
    a.reset_env = (RS.env_actions != nullptr ? 1 : 0) | (RS.episode_length != nullptr ? 2 : 0);
    if (RS.env_actions) {
        UNSUP(a.env_actions && a.env_actions != RS.env_actions);
        UNSUP(!RS.env_last_actions || (a.env_last_actions && a.env_last_actions != RS.env_last_actions));
        a.env_actions = RS.env_actions; a.env_last_actions = RS.env_last_actions;
    }
    UNSUP(RS.episode_length && RS.episode_length != T.episode_length);
    UNSUP(RS.max_episode_length && T.max_episode_length && RS.max_episode_length != T.max_episode_length);
    if (RS.max_episode_length) a.max_episode_length = RS.max_episode_length;
    a.base_max_episode_length = RS.base_max_episode_length;
    a.max_random_scaling = RS.max_episode_length ? RS.max_random_scaling : 0.0f;
    a.n_air = RS.num_contact;
    for (int m = 0; m < RS.num_contact; ++m) {
        a.air_links[m] = RS.air_links[m];
        for (int s = 0; s < 4; ++s) a.air_state[m][s] = RS.air_state[m][s];
    }
    a.reset_dofs = RS.scene_dof_pos != nullptr;
    if (RS.scene_dof_pos) {
        UNSUP(a.dof_pos && a.dof_pos != RS.scene_dof_pos);
        UNSUP(!RS.default_dof_pos || (a.default_dof_pos && a.default_dof_pos != RS.default_dof_pos));
        a.dof_pos = RS.scene_dof_pos; a.default_dof_pos = RS.default_dof_pos; a.dof_vel = RS.scene_dof_vel;
        a.dof_noise_scale = RS.dof_noise_scale;
    }
    a.scene_reset = RS.scene_pos != nullptr;
    if (RS.scene_pos) {
        const bool writes_quat = RS.spawn_mode ? RS.spawn_set_quat != 0 : RS.set_quat != 0;
        GfEntityView ev{RS.scene_pos, writes_quat ? RS.scene_quat : nullptr, RS.zero_velocity ? RS.scene_lin_vel : nullptr,
                        RS.zero_velocity ? RS.scene_ang_vel : nullptr};
        GfEntityView cur{a.pos, a.quat, a.lin_vel, a.ang_vel};
        UNSUP(!same_entity(cur, ev));
        UNSUP(writes_quat && !RS.scene_quat);
        if (RS.spawn_mode) {
            UNSUP(RS.spawn_draws);  // dense parity draws run phase by phase
            UNSUP(RS.terrain.height_field && (RS.terrain.rows < 1 || RS.terrain.cols < 1));
            UNSUP(have_terrain && memcmp(&a.terrain, &RS.terrain, sizeof(GfTerrainView)) != 0);
            a.terrain = RS.terrain;
            have_terrain = true;
            a.spawn_mode = 1; a.spawn_set_quat = RS.spawn_set_quat; a.spawn_rot_mask = RS.spawn_rot_mask;
            a.spawn_x_min = RS.spawn_x_min; a.spawn_x_span = RS.spawn_x_span; a.spawn_y_min = RS.spawn_y_min; a.spawn_y_span = RS.spawn_y_span;
            a.spawn_height_offset = RS.spawn_height_offset;
            for (int j = 0; j < 3; ++j) { a.spawn_rot_lo[j] = RS.spawn_rot_lo[j]; a.spawn_rot_hi[j] = RS.spawn_rot_hi[j]; }
        }
        UNSUP(RS.zero_velocity && (!RS.scene_lin_vel || !RS.scene_ang_vel));
        merge_entity(a, ev);
        a.set_quat = RS.set_quat; a.zero_velocity = RS.zero_velocity; a.quat_stash = RS.quat_stash;
        // the stash takes the PRE-reset quaternion out of the control wave's registers: it has to be loaded even when no term or
        // observation item of THIS launch reads it (termination done by a launch of its own, the body-frame items in an observation
        // manager that runs behind this launch, a getter the training script calls between steps) — found by the fuzz soak, seeds 123 / 140
        if (RS.quat_stash && (RS.set_quat || (RS.spawn_mode && RS.spawn_set_quat))) needs |= PN_QUAT;
        for (int j = 0; j < 3; ++j) a.reset_pos[j] = RS.reset_pos[j];
        for (int j = 0; j < 4; ++j) a.reset_quat[j] = RS.reset_quat[j];
        if (RS.zero_velocity && RS.scene_dof_vel) {
            UNSUP(a.dof_vel && a.dof_vel != RS.scene_dof_vel);
            a.dof_vel = RS.scene_dof_vel;
        }
    }

    // observations
    a.n_obs = r->num_observe;
    int omax = 0, stale_seen = 0;
    for (int m = 0; m < a.n_obs; ++m) {
        const GfObservationArgs* ob = r->observe[m];
        UNSUP(!ob || ob->num_envs != N || ob->num_items > kPostMaxItems || ob->noise_draws || !ob->obs);
        UNSUP(ob->seed != RS.seed || ob->env_offset != RS.env_offset);
        UNSUP(ob->history_ring < 0 || (int64_t)ob->history_ring > (ob->ring_slots ? (int64_t)ob->ring_slots : (int64_t)ob->history_len));
        UNSUP(ob->history_len > 1 && !ob->history_ring && !ob->prev_obs);
        UNSUP((reinterpret_cast<uintptr_t>(ob->obs) & 15u) || (ob->prev_obs && (reinterpret_cast<uintptr_t>(ob->prev_obs) & 15u)));
        GfEntityView cur{a.pos, a.quat, a.lin_vel, a.ang_vel};
        PostObs& po = a.obs[m];
        po.obs = ob->obs; po.prev = (ob->history_len > 1 && !ob->history_ring) ? ob->prev_obs : nullptr; po.stream = ob->stream;
        po.ring = ob->history_ring;
        po.ring_slots = (int32_t)ob->ring_slots;
        UNSUP(ob->ring_slots && (!ob->history_ring || (int64_t)ob->ring_slots < ob->history_len || (int64_t)ob->history_ring > (int64_t)ob->ring_slots));
        // (the frame of an in-place ring is read back by the gather that follows: cached)
        if (!po.ring && N * (int64_t)ob->obs_width * (ob->history_len > 0 ? ob->history_len : 1) * 4 >= kObsStreamBytes) a.obs_stream |= 1u << m;
        po.num_items = ob->num_items; po.width = ob->obs_width; po.history = ob->history_len;
        omax = ob->obs_width > omax ? ob->obs_width : omax;
        bool uses_entity = false;
        for (int i = 0; i < ob->num_items; ++i) {
            GfObsItem it = ob->items[i];
            switch (it.op) {
                case GF_O_COMMAND: {
                    UNSUP(it.i0 < 0 || it.i0 >= GF_MAX_COMMAND_VIEWS || !ob->command[it.i0].command || it.width != ob->command[it.i0].width);
                    const int s = pk.view_slot(ob->command[it.i0]);
                    UNSUP(s < 0);
                    UNSUP(pk.a.cmd_of_view[s] >= 0 && pk.a.cmd_of_view[s] != kViewGait && it.width > kPostMaxRanges);
                    it.i0 = s;
                } break;
                case GF_O_ANG_VEL_BODY: needs |= PN_QUAT | PN_ANG; uses_entity = true; break;
                case GF_O_LIN_VEL_BODY: needs |= PN_QUAT | PN_LIN; uses_entity = true; break;
                case GF_O_PROJ_GRAVITY: needs |= PN_QUAT; uses_entity = true; break;
                case GF_O_DOF_POS: needs |= PN_DOFPOS; UNSUP(!ob->dof_pos || (a.dof_pos && a.dof_pos != ob->dof_pos)); a.dof_pos = const_cast<float*>(ob->dof_pos); D = ob->num_dofs; break;
                case GF_O_DOF_VEL: needs |= PN_DOFVEL; UNSUP(!ob->dof_vel || (a.dof_vel && a.dof_vel != ob->dof_vel)); a.dof_vel = const_cast<float*>(ob->dof_vel); D = ob->num_dofs; break;
                case GF_O_DOF_FORCE: UNSUP(!ob->dof_force || (a.dof_force && a.dof_force != ob->dof_force)); a.dof_force = ob->dof_force; D = ob->num_dofs; break;
                case GF_O_ACTIONS: needs |= PN_TARGETS; UNSUP(!ob->targets || (a.targets && a.targets != ob->targets)); a.targets = ob->targets; D = ob->num_dofs; break;
                case GF_O_RAW_ACTIONS: needs |= PN_ACTIONS; UNSUP(!ob->env_actions || (a.env_actions && a.env_actions != ob->env_actions)); a.env_actions = const_cast<float*>(ob->env_actions); D = ob->num_dofs; break;
                case GF_O_CONTACT_FORCE_NORM: {
                    UNSUP(it.i0 < 0 || it.i0 >= GF_MAX_CONTACT_VIEWS || !ob->contact[it.i0].contacts || it.width != ob->contact[it.i0].num_links);
                    const int s = pk.contact_slot(ob->contact[it.i0]);
                    UNSUP(s < 0);
                    it.i0 = s;
                } break;
                default: return GF_E_UNSUPPORTED;
            }
            if (it.op == GF_O_DOF_POS || it.op == GF_O_DOF_VEL || it.op == GF_O_ACTIONS || it.op == GF_O_RAW_ACTIONS || it.op == GF_O_DOF_FORCE)
                UNSUP(it.width != ob->num_dofs);
            po.items[i] = it;
        }
        if (uses_entity) {
            UNSUP(!same_entity(cur, ob->entity));
            merge_entity(a, ob->entity);
            // stale quaternion source must be this step's reset (or absent when the reset does not touch quat)
            if (obs_only) {
                // the reset ran (or, in a step without a done env, did not run) as a launch of its own: the manager's descriptor says
                // whether this tick has a stash to read, and every manager of the launch must say the same
                UNSUP(ob->stale_quat && (ob->stale_quat != RS.quat_stash || ob->stale_mask != T.terminated || ob->stale_mask2 != T.truncated));
                stale_seen |= ob->stale_quat ? 1 : 2;
            } else if (a.scene_reset && (a.spawn_mode ? a.spawn_set_quat : a.set_quat)) {
                UNSUP(ob->stale_quat != a.quat_stash || ob->stale_mask != T.terminated || ob->stale_mask2 != T.truncated);
            } else {
                UNSUP(ob->stale_quat != nullptr);
            }
        }
    }
    if (obs_only) {
        UNSUP(stale_seen == 3);
        a.quat_stash = (stale_seen & 1) ? RS.quat_stash : nullptr;
    }
    UNSUP(omax >= GF_MAX_OBS_WIDTH);

    // rollout-storage rows (§8f-5): second stores of what the launch holds anyway
    if (const GfRolloutArgs* ro = r->rollout) {
        UNSUP(ro->num_envs != N);
        UNSUP((ro->done_out && (ro->terminated != T.terminated || ro->truncated != T.truncated)) || (ro->reward_out && (!RW || ro->reward != RW->reward)));
        a.roll_reward = ro->reward_out; a.roll_done = ro->done_out;
        a.roll_obs = nullptr; a.roll_obs_index = -1;
        if (ro->obs_out) {
            for (int m = 0; m < a.n_obs; ++m)
                if (a.obs[m].obs == ro->obs && a.obs[m].width * a.obs[m].history == ro->obs_width) a.roll_obs_index = m;
            UNSUP(a.roll_obs_index < 0 || (reinterpret_cast<uintptr_t>(ro->obs_out) & 15u));
            UNSUP(a.obs[a.roll_obs_index].ring != 0);   // an in-place ring is not a newest-first row: nothing to copy from
            a.roll_obs = ro->obs_out;
        }
    }

    // everything the kernel dereferences must exist, be float4-aligned and agree on D
    a.num_dofs = D > 0 ? D : 4;   // (no phase of the launch reads a DOF row — possible in front of a reset that user code runs: one unused chunk)
    a.needs = needs;
    UNSUP((needs & PN_QUAT) && !a.quat);
    UNSUP((needs & PN_POS) && !a.pos);
    UNSUP((needs & PN_LIN) && !a.lin_vel);
    UNSUP((needs & PN_ANG) && !a.ang_vel);
    UNSUP((needs & PN_EPLEN) && !a.episode_length);
    // DOF rows are ceil(D / 4) float4 chunks (a last chunk of fewer than four floats is handled element by element, gf_post_args.h):
    // 12 and 28 DOF have static programs / both kernel variants, every other count up to 32 a four-wave interpreter variant
    const bool dofs_ok = D >= 1 && D <= 32;
    UNSUP((needs & (PN_DOFPOS | PN_DOFVEL | PN_TARGETS | PN_ACTIONS | PN_LAST)) && !dofs_ok);
    UNSUP((a.reset_dofs || (a.reset_env & 1)) && !dofs_ok);
    UNSUP((needs & PN_DOFPOS) && !a.dof_pos);
    UNSUP((needs & PN_DOFVEL) && !a.dof_vel);
    UNSUP((needs & PN_TARGETS) && !a.targets);
    UNSUP((needs & PN_ACTIONS) && !a.env_actions);
    UNSUP((needs & PN_LAST) && !a.env_last_actions);
    UNSUP((needs & PN_DOFDEV) && !a.default_dof_pos);
    const void* al[] = {a.quat, a.dof_pos, a.dof_vel, a.targets, a.env_actions, a.env_last_actions, a.default_dof_pos, a.quat_stash, a.dof_force};
    for (const void* p : al) UNSUP(reinterpret_cast<uintptr_t>(p) & 15u);
    UNSUP((int64_t)N * (D > 4 ? D : 4) * 4 >= (int64_t)1 << 32);
    return GF_OK;
}


// ---- the scene's ContactManagers in front of the other phases, in the same launch (GfPostArgs.cfold) ------------------------------
int validate_contact(const GfContactArgs* a);                                   // gf_contact.hip
bool contact_compatible(const GfContactArgs* x, const GfContactArgs* y);
ContactMgrL contact_mgr_image(const GfContactArgs* a);

// GF_OK: pk.a.cfold describes the managers and the launch runs their step first; GF_E_UNSUPPORTED: the caller launches the contact
// kernel itself (more than 32 tracked links / 16 with-filter links / 255 scene links, slot rows that do not fit the tile's LDS,
// a statistics block other than the step's, an observation-only launch, GF_OPT_FOLD_CONTACT = 0).
static int fold_contacts(Packer& pk, const GfContactArgs* const* mgrs, int num) {
    GfPostArgs& a = pk.a;
    UNSUP(!g_options[GF_OPT_FOLD_CONTACT] || g_options[GF_OPT_POST_VARIANT] == 0);
    UNSUP(num < 1 || num > kFoldMaxMgr || a.obs_only);
    PostContact& f = a.cfold;
    int total = 0;
    for (int m = 0; m < num; ++m) {
        const GfContactArgs* c = mgrs[m];
        const int rc = validate_contact(c);
        if (rc) return rc;
        UNSUP(m > 0 && !contact_compatible(mgrs[0], c));
        UNSUP(c->num_envs != a.num_envs || c->num_contacts < 1 || c->num_scene_links > 255);
        UNSUP(total + c->num_targets > kFoldMaxTargets || c->num_with > kFoldMaxWith);
        UNSUP(c->stats && c->stats != a.stats);
        const ContactMgrL l = contact_mgr_image(c);
        PostContactMgr& o = f.m[m];
        o.contacts = l.contacts; o.contact_positions = l.contact_positions; o.position_counts = l.position_counts;
        o.link_vel_out = l.link_vel_out; o.link_pos_out = l.link_pos_out;
        o.last_air_time = l.last_air_time; o.current_air_time = l.current_air_time;
        o.last_contact_time = l.last_contact_time; o.current_contact_time = l.current_contact_time;
        o.air_time_threshold = l.air_time_threshold;
        o.num_targets = (uint8_t)l.num_targets; o.num_with = (uint8_t)l.num_with;
        o.has_with_filter = l.has_with_filter ? 1 : 0; o.track_air_time = l.track_air_time ? 1 : 0;
        for (int w = 0; w < c->num_with; ++w) {
            UNSUP(c->with_link_ids[w] < 0 || c->with_link_ids[w] > 254);
            o.with_ids[w] = (uint8_t)c->with_link_ids[w];
        }
        for (int t = 0; t < c->num_targets; ++t) {
            UNSUP(c->target_link_ids[t] < 0 || c->target_link_ids[t] > 254);
            f.target_ids[total] = (uint8_t)c->target_link_ids[t];
            f.mgr_of[total] = (uint8_t)m;
            f.local_of[total] = (uint8_t)t;
            ++total;
        }
    }
    const GfContactArgs* c0 = mgrs[0];
    UNSUP(fold_lds_bytes(c0->num_contacts) + sizeof(GfPostArgs) > 60 * 1024);
    // Measured on one box with the fold as shipped and switched off (tools/scaling_table.sh, profiles/r04_t_scaling.jsonl; us per step):
    //   one pass of the tile's 256 lanes (up to 4 tracked links: humanoid 3, contacts 4)  8 192 … 1 024 envs: 24.7 / 23.1 / 22.8 / 22.6 vs 26.6 / 26.6 / 25.3 / 25.0
    //   three passes (rough terrain, 9 links)   16 384 / 8 192 / 4 096 / 2 048 envs: 32.0 / 27.4 / 25.1 / 25.1 vs 31.6 / 25.5 / 25.8 / 25.2
    //   four passes (gait task, 13 links)       65 536 / 32 768 / 16 384 / 8 192 envs: 118.2 / 73.4 / 51.0 / 39.9 vs 125.5 / 77.2 / 50.3 / 39.3
    // A tile that needs several passes pays them one after the other on ONE CU, where a launch of its own spreads the same pairs over
    // four times as many small workgroups: below these sizes (128-256 tiles on 256 CUs) that costs more than the launch it saves.
    // GF_OPT_FOLD_CONTACT = 2 folds regardless.
    const int passes = (total * kEnvBlock + kWsBlock - 1) / kWsBlock;
    const int min_envs = passes >= 4 ? 32768 : (passes >= 2 ? 16384 : 0);
    UNSUP(g_options[GF_OPT_FOLD_CONTACT] != 2 && a.num_envs < min_envs);
    f.force = c0->force; f.position = c0->position; f.links_quat = c0->links_quat; f.links_vel = c0->links_vel; f.links_pos = c0->links_pos;
    f.link_a = c0->link_a; f.link_b = c0->link_b;
    f.num_contacts = c0->num_contacts; f.num_scene_links = c0->num_scene_links; f.num_mgr = num; f.total_targets = total;
    f.dt = c0->dt;
    return GF_OK;
}

static int post_launch(const GfPostArgs& a, hipStream_t s);
bool post_program_folds(const GfPostArgs& a);   // the kernel this descriptor selects carries the contact phase

int post_step(const GfPostRefs* r, const GfContactArgs* const* mgrs, int num_mgr, hipStream_t s) {
    Packer pk;
    int rc = pack(r, pk);
    if (rc) return rc;
    if (num_mgr > 0) {
        UNSUP(!post_program_folds(pk.a));
        rc = fold_contacts(pk, mgrs, num_mgr);
        if (rc) return rc;
    }
    return post_launch(pk.a, s);
}

}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_post_physics_check(const GfPostRefs* r) {
    gf::Packer pk;
    return gf::pack(r, pk);
}

using gf::lds_ws_floats;

// ---- programs compiled at run time (include/gf_step.h: gf_post_program_register) ---------------------------------------------
// A plugin is a small shared object built from csrc/gf_post_ws.h + ONE generated program struct (genesis_forge_amd/_programs.py):
// it carries post_ws_kernel<P> in its own code object and exports the matcher, the kernel's host handle and its LDS size.  The
// library only keeps the table; ids start at kDynBase.  Registration is rare and append-only (fixed-size table, the count is
// published last), selection walks it on every launch of a config no built-in program matches.
namespace {
constexpr int kDynBase = 100, kDynMax = 256;   // (programs registered per process)
struct DynProgram {
    void* dl;
    char name[64];
    int (*matches)(const gf::GfPostArgs*);
    const void* kernel;
    size_t (*lds_bytes)(int, int);
    bool folds;   // gfp_folds(): the plugin's kernel carries the contact phase (ws_prog_folds<P>)
};
DynProgram g_dyn[kDynMax];
std::atomic<int> g_dyn_count{0};
std::mutex g_dyn_mutex;
}  // namespace

extern "C" __attribute__((visibility("default"))) int gf_post_program_register(const char* path, int* id_out) {
    if (!path) return GF_E_NULL;
    void* dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!dl) return GF_E_UNSUPPORTED;
    auto abi = (int (*)(void))dlsym(dl, "gfp_abi_version");
    auto asz = (int (*)(void))dlsym(dl, "gfp_args_size");
    auto nm = (const char* (*)(void))dlsym(dl, "gfp_name");
    auto mt = (int (*)(const gf::GfPostArgs*))dlsym(dl, "gfp_matches");
    auto kn = (const void* (*)(void))dlsym(dl, "gfp_kernel");
    auto ld = (size_t (*)(int, int))dlsym(dl, "gfp_lds_bytes");
    // the packed descriptor is the interface between library and plugin: both must come from the same headers
    if (!abi || !asz || !nm || !mt || !kn || !ld || abi() != GF_ABI_VERSION || asz() != (int)sizeof(gf::GfPostArgs)) {
        dlclose(dl);
        return GF_E_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> lock(g_dyn_mutex);
    const int n = g_dyn_count.load();
    for (int i = 0; i < n; ++i)
        if (!strncmp(g_dyn[i].name, nm(), sizeof(g_dyn[i].name) - 1)) {   // already there (same signature hash in the name)
            dlclose(dl);
            if (id_out) *id_out = kDynBase + i;
            return GF_OK;
        }
    if (n >= kDynMax) { dlclose(dl); return GF_E_RANGE; }
    DynProgram& d = g_dyn[n];
    d.dl = dl;
    snprintf(d.name, sizeof(d.name), "%s", nm());
    d.matches = mt;
    d.kernel = kn();
    d.lds_bytes = ld;
    auto fo = (int (*)(void))dlsym(dl, "gfp_folds");
    d.folds = fo && fo() != 0;
    g_dyn_count.store(n + 1);
    if (id_out) *id_out = kDynBase + n;
    return GF_OK;
}

extern "C" __attribute__((visibility("default"))) int gf_post_program_count(void) { return g_dyn_count.load(); }

// Static programs in registration order; program id = 1 + index (0 = table interpreter).
#define GF_POST_PROGRAMS(X)                                                                                                \
    X(1, gf::ProgGo2CommandDirection) X(2, gf::ProgGo2Simple) X(3, gf::ProgGo2Contacts) X(4, gf::ProgGo2RoughTerrain) \
    X(5, gf::ProgBerkeleyHumanoid) X(6, gf::ProgGo2GaitTrainer) \
    X(7, gf::ProgHumanoid28Stress) X(8, gf::ProgGo2GaitTrainerFront) X(9, gf::ProgGo2GaitTrainerObs)

static int select_program(const gf::GfPostArgs& a) {
    if (gf::g_options[GF_OPT_POST_VARIANT] < 2) return 0;
#define GF_MATCH(id, P) \
    if (gf::program_matches<P>(a)) return id;
    GF_POST_PROGRAMS(GF_MATCH)
#undef GF_MATCH
    const int n = g_dyn_count.load();
    for (int i = 0; i < n; ++i)
        if (g_dyn[i].matches(&a)) return kDynBase + i;
    return 0;
}

bool gf::post_program_folds(const gf::GfPostArgs& a) {
    const int id = select_program(a);
    if (id >= kDynBase) return g_dyn[id - kDynBase].folds;
#define GF_FOLDS(pid, P) \
    if (id == pid) return gf::ws_prog_folds<P>();
    GF_POST_PROGRAMS(GF_FOLDS)
#undef GF_FOLDS
    return true;   // the table interpreter
}

extern "C" __attribute__((visibility("default"))) int gf_post_physics_describe(const GfPostRefs* r, char* buf, int cap) {
    gf::Packer pk;
    const int rc = gf::pack(r, pk);
    if (rc) return rc;
    if (!buf || cap <= 0) return GF_E_NULL;
    const int id = select_program(pk.a);
    const char* name = "interpreter";
#define GF_NAME(pid, P) \
    if (id == pid) name = P::name;
    GF_POST_PROGRAMS(GF_NAME)
#undef GF_NAME
    if (id >= kDynBase) name = g_dyn[id - kDynBase].name;
    int n = snprintf(buf, (size_t)cap, "program %d (%s): ", id, name);
    if (n < cap) gf::describe_program(pk.a, buf + n, cap - n);
    return GF_OK;
}

int gf::post_launch(const gf::GfPostArgs& packed, hipStream_t s) {
#ifdef GF_STAMPS
    gf::GfPostArgs stamped = packed;
    stamped.stamps = gf_debug_stamps;
    stamped.stamp_block = (uint32_t)(packed.num_envs / 64 / 2);
    const gf::GfPostArgs& a = stamped;
#else
    const gf::GfPostArgs& a = packed;
#endif
    int omax = 0;
    for (int m = 0; m < a.n_obs; ++m) omax = a.obs[m].width > omax ? a.obs[m].width : omax;
    const size_t lds = sizeof(gf::GfPostArgs) + ((size_t)(gf::kPostMaxReward + gf::kPostAuxRows) * gf::kEnvBlock + (size_t)(omax + 1) * gf::kEnvBlock) * sizeof(float);
    // a folded contact phase stages the tile's slot ids in the LDS the later phases use (behind the interpreter's descriptor copy)
    const size_t fold_lds = a.cfold.num_mgr > 0 ? gf::fold_lds_bytes(a.cfold.num_contacts) : 0;
    auto with_fold = [&](size_t bytes, bool interp) {
        const size_t need = fold_lds ? fold_lds + (interp ? sizeof(gf::GfPostArgs) : 0) : 0;
        return (bytes > need ? bytes : need) + (size_t)gf::kWsTilesLdsFloats * sizeof(float);
    };
    const unsigned grid1 = gf::env_grid(a.num_envs);
    gf::PhaseScope scope(GF_PHASE_POST, s);
    bool any_ring = false;
    for (int m = 0; m < a.n_obs; ++m) any_ring = any_ring || a.obs[m].ring != 0;
    const bool ws_only = a.n_gait || a.roll_obs || a.roll_reward || a.roll_done || any_ring || a.term_done || a.no_reset || (a.num_dofs != 12 && a.num_dofs != 28);   // the one-wave variant has neither a gait manager nor rollout stores
    const unsigned grid = grid1;
    if (gf::g_options[GF_OPT_POST_VARIANT] == 0 && !ws_only) {
        if (a.num_dofs == 28) GF_LAUNCH(scope, gf::post_kernel<7>, grid1, gf::kEnvBlock, lds, s, a);
        else GF_LAUNCH(scope, gf::post_kernel<3>, grid1, gf::kEnvBlock, lds, s, a);
    } else if (const int prog = select_program(a); prog >= kDynBase) {
        // a program compiled at run time: the plugin's kernel handle, launched like any other (launch sink, dispatch events)
        const DynProgram& d = g_dyn[prog - kDynBase];
        const size_t lds_dyn = with_fold(d.lds_bytes(omax, a.n_gait), false);
        void* kargs[1] = {const_cast<gf::GfPostArgs*>(&a)};
        if (scope.active()) {
            scope.use_dispatch_events();
            (void)hipExtLaunchKernel(d.kernel, dim3(grid1), dim3(gf::kWsBlock), kargs, lds_dyn, s, scope.start(), scope.stop(), 0);
        } else {
            gf::sink_launch(d.kernel, dim3(grid1), dim3(gf::kWsBlock), lds_dyn, s, kargs);
        }
    } else if (prog) {
#define GF_RUN(id, P) \
        if (prog == id) GF_LAUNCH(scope, gf::post_ws_kernel<P>, grid, gf::kWsBlock, with_fold(lds_ws_floats<P>(omax, a.n_gait) * sizeof(float), false), s, a);
        GF_POST_PROGRAMS(GF_RUN)
#undef GF_RUN
    } else {
#define GF_RUN_INTERP_T(DV_, T_)                                                                                              \
        do {                                                                                                                  \
            using Var = gf::Interp<DV_, T_>;                                                                                  \
            const size_t lds_var = with_fold(sizeof(gf::GfPostArgs) + lds_ws_floats<Var>(omax, a.n_gait) * sizeof(float), true); \
            GF_LAUNCH(scope, gf::post_ws_kernel<Var>, grid, gf::kWsBlock, lds_var, s, a);                                     \
        } while (0)
#define GF_RUN_INTERP(DV_) \
        do { if (a.num_dofs == 4 * DV_) GF_RUN_INTERP_T(DV_, false); else GF_RUN_INTERP_T(DV_, true); } while (0)
        switch ((a.num_dofs + 3) / 4) {   // chunks per row
            case 1: GF_RUN_INTERP(1); break;
            case 2: GF_RUN_INTERP(2); break;
            case 4: GF_RUN_INTERP(4); break;
            case 5: GF_RUN_INTERP(5); break;
            case 6: GF_RUN_INTERP(6); break;
            case 7: GF_RUN_INTERP(7); break;
            case 8: GF_RUN_INTERP(8); break;
            default:   // 9 … 12 DOF; configs without DOF rows (num_dofs of the action manager all the same; pack() let nothing above 32 through)
                if (a.num_dofs >= 9 && a.num_dofs < 12) GF_RUN_INTERP_T(3, true); else GF_RUN_INTERP_T(3, false);
                break;
        }
#undef GF_RUN_INTERP_T
#undef GF_RUN_INTERP
    }
    return gf::launch_status();
}

extern "C" __attribute__((visibility("default"))) int gf_post_physics_step(const GfPostRefs* r, void* stream) {
    return gf::post_step(r, nullptr, 0, (hipStream_t)stream);
}

extern "C" __attribute__((visibility("default"))) int gf_post_physics_step_contacts(const GfPostRefs* r, const GfContactArgs* const* contacts, int num_contacts,
                                                                                   void* stream) {
    if (num_contacts < 0 || (num_contacts > 0 && !contacts)) return GF_E_NULL;
    return gf::post_step(r, contacts, num_contacts, (hipStream_t)stream);
}
