// gf_reward.hip — Phase B4: RewardManager.step as ONE fused launch (the north-star kernel).
//
// Replaces managers/reward_manager.py:166-195 plus every term body of mdp/rewards.py
// (73 aten ops / 60 launches for the 6-term Go2 config in the reference).
//
// Mapping: one lane per env, one wave per workgroup.  Every input a term can need is requested at
// the top of the kernel — quat as one dwordx4, the [N,D] rows as D/4 dwordx4 per lane (a wave covers a
// contiguous 64*D*4-byte span, so every fetched line is fully used), [N,3] vectors as 3 dwords — so
// a wave has ≈17 KB in flight before its first wait; with 4 waves per CU at N=65536 that is the
// ≈70 KB/CU the HBM latency needs.  The body-frame rotations (the reference recomputes them in
// every term that asks, entity_manager.py:130-146) are computed once per env in registers.
//
// The term table lives in the kernarg segment and is walked with wave-uniform scalar loads: the
// per-term switch is a scalar branch, never lane divergence.  Per-term episode sums are SoA [T,N]
// (each term column coalesced).  Their read-modify-write would otherwise serialise T memory round
// trips inside the term loop, so the T columns are first pulled into LDS with direct-to-LDS loads
// (global_load_lds_dword: no VGPR, no wait) issued back to back ahead of the arithmetic; the term
// loop then reads its own lane's slot from LDS, adds, and streams the new sum out.
//
// Accumulation order is the reference's: left fold over the cfg order, each term multiplied by
// (float)(weight*dt) first (reward_manager.py:185-189); zero-weight terms never enter the table.
//
// Algorithmic traffic, Go2 command config (T=6, D=12):  R pos 12 + quat 16 + vel 12 + ang 12 +
// dof_pos 48 + actions 48 + last_actions 48 + cmd 12 = 208;  RW episode_sums 8T = 48;  RW secs 8;
// W reward 4  →  268 B/env  (SURVEY.md §8d).
#include "gf_launch.h"
#include "gf_terms.h"

namespace gf {

enum : uint32_t {
    RN_POS = 1, RN_QUAT = 2, RN_LIN = 4, RN_ANG = 8, RN_GRAV = 16,
    RN_DOF_DEV = 32, RN_ACT_RATE = 64, RN_TERMINATED = 128, RN_CMD0 = 256,
};

// the phase for the 64 envs of workgroup blockIdx.x; lanes = threadIdx.x < 64; lds_sums: GF_MAX_TERMS * 64 floats of LDS
template <int DV>
__device__ __forceinline__ void reward_body(const GfRewardArgs& a, const uint32_t needs, float* lds_sums) {

    const int64_t N = a.num_envs;
    const int64_t n_raw = (int64_t)blockIdx.x * kEnvBlock + threadIdx.x;
    const bool live = n_raw < N;
    const int64_t n = live ? n_raw : N - 1;  // tail lanes shadow the last env, stores are masked
    const int D = a.num_dofs;
    const int T = a.num_terms;
    const bool step_mode = a.mode == GF_REWARD_MODE_STEP;
    const bool logging = step_mode && a.logging_enabled;

    // ---- 1. episode-sum columns: global -> LDS, asynchronously (no VGPR, no wait) -------------
    if (logging) {
        for (int k = 0; k < T; ++k) {
            const float* src = a.episode_sums + (int64_t)a.terms[k].row * N + n;
            __builtin_amdgcn_global_load_lds(src, lds_sums + k * kEnvBlock, 4, 0, 0);
        }
    }

    // ---- 2. every per-env input, all loads issued before the first use -------------------------
    // Inputs a config does not need are redirected to a zero pad inside the code object instead of
    // being skipped with a branch: the load stream stays straight-line (no phi copies, no per-block
    // waits), and an unneeded field costs one broadcast L2 hit per wave instead of HBM traffic.
    const bool nq = needs & RN_QUAT, np = needs & RN_POS, nl = needs & RN_LIN, na = needs & RN_ANG;
    const bool nd = needs & RN_DOF_DEV, nr = needs & RN_ACT_RATE, nt = needs & RN_TERMINATED, nc = needs & RN_CMD0;
    // Pull every pointer out of the kernarg segment in one batch of scalar loads (pinned in SGPRs)
    // instead of one dependent s_load + wait in front of each vector load.
    const float *k_quat = a.entity.quat, *k_pos = a.entity.pos, *k_lin = a.entity.lin_vel, *k_ang = a.entity.ang_vel;
    const float *k_dof = a.dof_pos, *k_def = a.default_dof_pos, *k_act = a.actions, *k_last = a.last_actions;
    const float *k_secs = a.episode_seconds, *k_cmd = a.command[0].command;
    const uint8_t* k_term = a.terminated;
    int k_cw = a.command[0].stride ? a.command[0].stride : a.command[0].width;
    asm volatile("" : "+s"(k_quat), "+s"(k_pos), "+s"(k_lin), "+s"(k_ang), "+s"(k_dof), "+s"(k_def));
    asm volatile("" : "+s"(k_act), "+s"(k_last), "+s"(k_secs), "+s"(k_cmd), "+s"(k_term), "+s"(k_cw));

    const uint32_t e = (uint32_t)n;  // 32-bit element offsets: SGPR base + VGPR offset addressing
    const float4 q = ldg4(gsel(nq, k_quat, 4u * e));
    const GF_GLOBAL float* pp = gsel(np, k_pos, 3u * e);
    const GF_GLOBAL float* lp = gsel(nl, k_lin, 3u * e);
    const GF_GLOBAL float* ap = gsel(na, k_ang, 3u * e);
    const V3 pos{pp[0], pp[1], pp[2]}, lin{lp[0], lp[1], lp[2]}, ang{ap[0], ap[1], ap[2]};
    const float secs_in = *gsel(step_mode, k_secs, e);
    const int terminated = *gsel(nt, k_term, e);
    // command view 0 (the VelocityCommandManager in every example): first 3 columns
    const uint32_t cw = nc ? (uint32_t)k_cw : 0u;
    const GF_GLOBAL float* cp = gsel(nc, k_cmd, e * cw);
    const float cmd0[3] = {cp[0], cp[cw > 1 ? 1 : 0], cp[cw > 2 ? 2 : 0]};

    float dof_dev = 0.f, act_rate = 0.f;
    if (DV > 0) {
        constexpr int R = DV > 0 ? DV : 1;
        float4 rp[R], ra[R], rl[R], df[R];
        const uint32_t ro = e * (uint32_t)D;
        const GF_GLOBAL float* p_pos = gsel(nd, k_dof, ro);
        const GF_GLOBAL float* p_def = gsel(nd, k_def, 0u);
        const GF_GLOBAL float* p_last = gsel(nr, k_last, ro);
        const GF_GLOBAL float* p_act = gsel(nr, k_act, ro);
#pragma unroll
        for (int c = 0; c < DV; ++c) { rp[c] = ldg4(p_pos + 4 * c); rl[c] = ldg4(p_last + 4 * c); ra[c] = ldg4(p_act + 4 * c); df[c] = ldg4(p_def + 4 * c); }
#pragma unroll
        for (int c = 0; c < DV; ++c) {
            dof_dev += fabsf(rp[c].x - df[c].x);
            dof_dev += fabsf(rp[c].y - df[c].y);
            dof_dev += fabsf(rp[c].z - df[c].z);
            dof_dev += fabsf(rp[c].w - df[c].w);
        }
#pragma unroll
        for (int c = 0; c < DV; ++c) {
            float d;
            d = rl[c].x - ra[c].x; act_rate += d * d;
            d = rl[c].y - ra[c].y; act_rate += d * d;
            d = rl[c].z - ra[c].z; act_rate += d * d;
            d = rl[c].w - ra[c].w; act_rate += d * d;
        }
    } else {
        if (nd)
            for (int d = 0; d < D; ++d) dof_dev += fabsf(a.dof_pos[n * D + d] - a.default_dof_pos[d]);
        if (nr)
            for (int d = 0; d < D; ++d) {
                const float df = a.last_actions[n * D + d] - a.actions[n * D + d];
                act_rate += df * df;
            }
    }

    // ---- 3. body-frame quantities, once per env -------------------------------------------------
    const V3 blin = rot_inv(q, lin);
    const V3 bang = rot_inv(q, ang);
    const V3 grav = rot_inv(q, V3{0.f, 0.f, -1.f});

    // the LDS-DMA columns must have landed before the first ds_read below; nothing but this wave's
    // vmcnt orders a ds_read behind a pending direct-to-LDS load.
    if (logging) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- 4. term loop (wave-uniform control flow); term bodies live in gf_terms.h ---------------
    RewardRegs rr;
    rr.pos = pos; rr.blin = blin; rr.bang = bang; rr.grav = grav; rr.dof_dev = dof_dev; rr.act_rate = act_rate;
    rr.terminated = terminated; rr.cmd0[0] = cmd0[0]; rr.cmd0[1] = cmd0[1]; rr.cmd0[2] = cmd0[2]; rr.n = n; rr.live = live;
    float buf = 0.f;
    for (int k = 0; k < T; ++k) {
        const GfTerm& t = a.terms[k];
        float v = 0.f;
        v = eval_reward_term(t, a, rr);
        if (!step_mode) {
            if (live) a.term_out[(int64_t)t.row * N + n_raw] = v;
            continue;
        }
        v = v * t.w;   // fn(...) * (weight*dt)        reward_manager.py:185-186
        buf += v;      // self._reward_buf += value    reward_manager.py:189
        if (logging) { // self._episode_data[name] += value   reward_manager.py:192-193
            const float s = lds_sums[k * kEnvBlock + threadIdx.x] + v;
            if (live) a.episode_sums[(int64_t)t.row * N + n_raw] = s;
        }
    }
    if (step_mode && live) {
        a.reward[n_raw] = buf;
        a.episode_seconds[n_raw] = secs_in + a.dt;  // self._episode_seconds += dt   reward_manager.py:178
    }
}

#ifndef GF_BODIES_ONLY
template <int DV>
__global__ __launch_bounds__(kEnvBlock) void reward_kernel(const GfRewardArgs a, const uint32_t needs) {
    prefetch_args<GfRewardArgs>();
    __shared__ float lds_sums[GF_MAX_TERMS * kEnvBlock];
    reward_body<DV>(a, needs, lds_sums);
}
#endif

}  // namespace gf

#ifndef GF_BODIES_ONLY
namespace gf {
// validation, the inputs the term table needs, and the row-vector width the [N,D] rows allow (0 = scalar path)
int reward_prep(const GfRewardArgs* a, uint32_t* needs_out, int* dv_out) {
    if (!a) return GF_E_NULL;
    if (a->num_terms < 0 || a->num_terms > GF_MAX_TERMS || a->num_envs < 0 || a->num_dofs < 0) return GF_E_RANGE;
    if (a->mode == GF_REWARD_MODE_STEP) {
        if (!a->reward || !a->episode_seconds) return GF_E_NULL;
        if (a->logging_enabled && !a->episode_sums) return GF_E_NULL;
    } else if (a->mode == GF_REWARD_MODE_EVAL) {
        if (!a->term_out) return GF_E_NULL;
    } else {
        return GF_E_RANGE;
    }
    uint32_t needs = 0;
    auto need_cmd = [&](int idx, int min_width) -> int {
        if (idx < 0 || idx >= GF_MAX_COMMAND_VIEWS || !a->command[idx].command) return GF_E_SLOT;
        return a->command[idx].width >= min_width ? GF_OK : GF_E_RANGE;
    };
    auto need_contact = [&](int idx) -> int {
        if (idx < 0 || idx >= GF_MAX_CONTACT_VIEWS || !a->contact[idx].contacts) return GF_E_SLOT;
        return a->contact[idx].num_links > 0 ? GF_OK : GF_E_RANGE;
    };
    for (int k = 0; k < a->num_terms; ++k) {
        const GfTerm& t = a->terms[k];
        if (t.row < 0 || t.row >= GF_MAX_TERMS) return GF_E_RANGE;
        int rc = GF_OK;
        switch (t.op) {
            case GF_R_IS_ALIVE:
            case GF_R_TERMINATED: needs |= gf::RN_TERMINATED; break;
            case GF_R_BASE_HEIGHT:
                needs |= gf::RN_POS;
                if (t.flags & GF_RW_FLAG_CMD) rc = need_cmd(t.i[0], 1);
                if ((t.flags & GF_RW_FLAG_TERRAIN) && a->terrain.height_field && (a->terrain.rows < 1 || a->terrain.cols < 1)) rc = GF_E_RANGE;
                break;
            case GF_R_DOF_SIMILAR_TO_DEFAULT: needs |= gf::RN_DOF_DEV; break;
            case GF_R_LIN_VEL_Z_L2: needs |= gf::RN_QUAT | gf::RN_LIN; break;
            case GF_R_ANG_VEL_XY_L2: needs |= gf::RN_QUAT | gf::RN_ANG; break;
            case GF_R_FLAT_ORIENTATION_L2: needs |= gf::RN_QUAT | gf::RN_GRAV; break;
            case GF_R_BODY_ACCEL_EXP:
                needs |= gf::RN_QUAT | gf::RN_LIN | gf::RN_ANG;
                if (t.i[0] < 0 || t.i[0] >= 4 || !a->state[t.i[0]]) rc = GF_E_SLOT;
                break;
            case GF_R_ACTION_RATE_L2: needs |= gf::RN_ACT_RATE; break;
            case GF_R_CMD_TRACK_LIN_VEL: needs |= gf::RN_QUAT | gf::RN_LIN; rc = need_cmd(t.i[0], 2); if (t.i[0] == 0) needs |= gf::RN_CMD0; break;
            case GF_R_CMD_TRACK_ANG_VEL: needs |= gf::RN_QUAT | gf::RN_ANG; rc = need_cmd(t.i[0], t.i[1] + 1); if (t.i[1] < 0) rc = GF_E_RANGE; if (t.i[0] == 0) needs |= gf::RN_CMD0; break;
            case GF_R_STAND_STILL: needs |= gf::RN_DOF_DEV; rc = need_cmd(t.i[0], 2); if (t.i[0] == 0) needs |= gf::RN_CMD0; break;
            case GF_R_HAS_CONTACT:
            case GF_R_CONTACT_FORCE: rc = need_contact(t.i[0]); break;
            case GF_R_FEET_AIR_TIME:
                rc = need_contact(t.i[0]);
                if (!rc && (!a->contact[t.i[0]].last_air_time || !a->contact[t.i[0]].current_contact_time)) rc = GF_E_SLOT;
                if (!rc && t.i[1] >= 0) { rc = need_cmd(t.i[1], 2); if (t.i[1] == 0) needs |= gf::RN_CMD0; }
                break;
            case GF_R_FEET_SLIDE:
                rc = need_contact(t.i[0]);
                if (!rc && !a->contact[t.i[0]].link_vel) rc = GF_E_SLOT;
                break;
            case GF_R_EXTERNAL:
                if (t.i[0] < 0 || t.i[0] >= GF_MAX_EXT || !a->ext[t.i[0]]) rc = GF_E_SLOT;
                break;
            case GF_R_GAIT_PHASE:
            case GF_R_FOOT_HEIGHT: {
                rc = need_contact(t.i[0]);
                if (!rc) rc = need_cmd(t.i[1], GF_GAIT_OBS_WIDTH);
                if (!rc && (a->command[t.i[1]].stride < GF_GAIT_ROW || !a->contact[t.i[0]].link_vel)) rc = GF_E_SLOT;
                if (!rc && t.op == GF_R_FOOT_HEIGHT && !a->contact[t.i[0]].link_pos) rc = GF_E_SLOT;
                for (int f = 0; !rc && f < 4; ++f)
                    if (((t.i[2] >> (8 * f)) & 0xff) >= a->contact[t.i[0]].num_links) rc = GF_E_RANGE;
            } break;
            default: return GF_E_OPCODE;
        }
        if (rc) return rc;
    }
    if ((needs & gf::RN_QUAT) && !a->entity.quat) return GF_E_NULL;
    if ((needs & gf::RN_POS) && !a->entity.pos) return GF_E_NULL;
    if ((needs & gf::RN_LIN) && !a->entity.lin_vel) return GF_E_NULL;
    if ((needs & gf::RN_ANG) && !a->entity.ang_vel) return GF_E_NULL;
    if ((needs & gf::RN_TERMINATED) && !a->terminated) return GF_E_NULL;
    if ((needs & gf::RN_DOF_DEV) && (!a->dof_pos || !a->default_dof_pos || a->num_dofs <= 0)) return GF_E_NULL;
    if ((needs & gf::RN_ACT_RATE) && (!a->actions || !a->last_actions || a->num_dofs <= 0)) return GF_E_NULL;
    if ((needs & gf::RN_QUAT) && (reinterpret_cast<uintptr_t>(a->entity.quat) & 15u)) return GF_E_UNSUPPORTED;
    auto al16 = [](const void* p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool rows16 = (a->num_dofs % 4 == 0) && al16(a->dof_pos) && al16(a->actions) && al16(a->last_actions) && al16(a->default_dof_pos);
    *needs_out = needs;
    *dv_out = (rows16 && a->num_dofs >= 8 && a->num_dofs <= 28) ? a->num_dofs / 4 : 0;
    return GF_OK;
}
}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_reward_step(const GfRewardArgs* a, void* stream) {
    uint32_t needs = 0;
    int dv = 0;
    const int rc = gf::reward_prep(a, &needs, &dv);
    if (rc) return rc;
    if (a->num_envs == 0) return GF_OK;
    const bool rows16 = dv > 0;
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = gf::env_grid(a->num_envs);
    gf::PhaseScope scope(GF_PHASE_REWARD, s);
    if (rows16 && a->num_dofs == 12) GF_LAUNCH(scope, gf::reward_kernel<3>, grid, gf::kEnvBlock, 0, s, *a, needs);
    else if (rows16 && a->num_dofs == 28) { scope.begin_bracket(); gf::klaunch(gf::reward_kernel<7>, dim3(grid), dim3(gf::kEnvBlock), 0, s, *a, needs); }
    else if (rows16 && a->num_dofs == 8) { scope.begin_bracket(); gf::klaunch(gf::reward_kernel<2>, dim3(grid), dim3(gf::kEnvBlock), 0, s, *a, needs); }
    else if (rows16 && a->num_dofs == 16) { scope.begin_bracket(); gf::klaunch(gf::reward_kernel<4>, dim3(grid), dim3(gf::kEnvBlock), 0, s, *a, needs); }
    else if (rows16 && a->num_dofs == 20) { scope.begin_bracket(); gf::klaunch(gf::reward_kernel<5>, dim3(grid), dim3(gf::kEnvBlock), 0, s, *a, needs); }
    else if (rows16 && a->num_dofs == 24) { scope.begin_bracket(); gf::klaunch(gf::reward_kernel<6>, dim3(grid), dim3(gf::kEnvBlock), 0, s, *a, needs); }
    else { scope.begin_bracket(); gf::klaunch(gf::reward_kernel<0>, dim3(grid), dim3(gf::kEnvBlock), 0, s, *a, needs); }
    return gf::launch_status();
}
#endif
