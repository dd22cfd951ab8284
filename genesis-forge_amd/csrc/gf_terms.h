// gf_terms.h — the term bodies of mdp/rewards.py and mdp/terminations.py as device functions, shared by the
// per-phase kernels (gf_reward.hip, gf_termination.hip) and the fused post-physics kernel (gf_post.hip) so the
// three can never drift apart.  `Args` is any descriptor exposing the members the bodies read
// (contact[], command[], ext[], state[], dt); per-env values arrive in registers.
#pragma once

#include "gf_device.h"

namespace gf {

struct TermRegs {
    int ep_len, max_len;
    bool has_maxlen;
    float tilt_sin;   // min(|g_xy|, 0.99) of the projected gravity
    V3 pos;
    int64_t m;        // env index (clamped for tail lanes)
};

// the contact-count terminations, given the count: one statement of the three rules for eval_termination_term below and for the
// fused kernel's control wave, which takes the counts early (their loads go out with the tile's other loads)
__host__ __device__ constexpr bool term_op_counts_contacts(int op) {
    return op == GF_T_HAS_CONTACT || op == GF_T_CONTACT_FORCE || op == GF_T_CONTACT_FORCE_GRACE;
}
__device__ __forceinline__ int termination_from_count(const GfTerm& t, int cnt, int ep_len) {
    switch (t.op) {
        case GF_T_HAS_CONTACT: return cnt >= t.i[1];
        case GF_T_CONTACT_FORCE: return cnt > 0;
        case GF_T_CONTACT_FORCE_GRACE: return !(ep_len <= t.i[1]) && (cnt > 0);
        default: return 0;
    }
}

template <class Args>
__device__ __forceinline__ int eval_termination_term(const GfTerm& t, const Args& a, const TermRegs& r, const uint32_t needs_maxlen) {
    const int ep_len = r.ep_len, max_len = r.max_len;
    const float tilt_sin = r.tilt_sin;
    const V3& pos = r.pos;
    const int64_t m = r.m;
    int v = 0;
    switch (t.op) {
        case GF_T_TIMEOUT:
            v = needs_maxlen ? (ep_len > max_len) : 0;
            break;
        case GF_T_BAD_ORIENTATION:
            // asin is monotone: asin(x) > radians(limit)  <=>  x > p0, where the host found p0 as
            // the largest f32 with asin(p0) <= (float)radians(limit) (same libm/torch asin as the
            // reference), so no device asinf can flip a mask.
            v = !(ep_len <= t.i[0]) && (tilt_sin > t.p[0]);
            break;
        case GF_T_BASE_HEIGHT_BELOW:
            v = pos.z < t.p[0];
            break;
        case GF_T_OUT_OF_BOUNDS:
            v = (pos.x < t.p[0]) || (pos.x > t.p[1]) || (pos.y < t.p[2]) || (pos.y > t.p[3]);
            break;
        case GF_T_HAS_CONTACT:
        case GF_T_CONTACT_FORCE:
        case GF_T_CONTACT_FORCE_GRACE:
            v = termination_from_count(t, contact_count_over(a.contact[t.i[0]], m, t.p[0]), ep_len);
            break;
        case GF_T_EXTERNAL:
            v = G(a.ext[t.i[0]])[m] != 0;
            break;
        default:
            break;
    }
    return v;
}

struct RewardRegs {
    V3 pos, blin, bang, grav;
    float dof_dev, act_rate;
    int terminated;
    float cmd0[3];    // first three columns of command view 0, before this step's resample
    int64_t n;
    bool live;
};

// the gait opcodes exist only behind descriptors that can carry them (GfRewardArgs here, GfPostArgs in gf_post_args.h)
template <class Args> struct HasGaitTerms { static constexpr bool value = false; };
template <> struct HasGaitTerms<GfRewardArgs> { static constexpr bool value = true; };

// ---- the memory-only reward terms in two halves ----------------------------------------------------------------------------------
// term_rows() requests what a term reads from memory into a block of registers (indexed with constants only: what a term does not
// use does not exist), term_value() turns the block into the term's value.  A kernel that evaluates several such terms calls every
// term_rows() first — all requests go out back to back and are waited for once — and the per-phase kernels call the two halves one
// after the other (eval_reward_term below): one definition of each term either way.  Terms over a run-time number of links hold the
// first four links in the block and load any further ones in the loop.
constexpr int kTermRowRegs = 32;

__host__ __device__ constexpr bool reward_op_has_rows(int op) {
    return op == GF_R_GAIT_PHASE || op == GF_R_FOOT_HEIGHT || op == GF_R_CONTACT_FORCE || op == GF_R_HAS_CONTACT || op == GF_R_FEET_SLIDE;
}

template <class Args>
__device__ __forceinline__ void term_rows(const GfTerm& t, const Args& a, int64_t n, float (&r)[kTermRowRegs]) {
    switch (t.op) {
        case GF_R_HAS_CONTACT:
        case GF_R_CONTACT_FORCE:
        case GF_R_FEET_SLIDE: {
            const GfContactView& cv = a.contact[t.i[0]];
            const int L = cv.num_links;
            if (L > 0) {
                float f[4][3];
                link_rows4(f, G(cv.contacts) + n * L * 3, 0, L);
#pragma unroll
                for (int j = 0; j < 4; ++j) { r[3 * j] = f[j][0]; r[3 * j + 1] = f[j][1]; r[3 * j + 2] = f[j][2]; }
                if (t.op == GF_R_FEET_SLIDE) {
                    link_rows4(f, G(cv.link_vel) + n * L * 3, 0, L);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { r[12 + 3 * j] = f[j][0]; r[13 + 3 * j] = f[j][1]; r[14 + 3 * j] = f[j][2]; }
                }
            }
        } break;
        case GF_R_GAIT_PHASE:
            if constexpr (HasGaitTerms<Args>::value) {   // phase, four offsets, per foot: contact force and link velocity
                const GfContactView& cv = a.contact[t.i[0]];
                const GfCommandView& gv = a.command[t.i[1]];
                const GF_GLOBAL float* g = G(gv.command) + n * cmd_stride(gv);
                const GF_GLOBAL float* fr = G(cv.contacts) + n * cv.num_links * 3;
                const GF_GLOBAL float* lv = G(cv.link_vel) + n * cv.num_links * 3;
                r[0] = g[GF_GAIT_PHASE];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int l = (t.i[2] >> (8 * f)) & 0xff;
                    r[1 + f] = g[GF_GAIT_OFFSET + f];
                    r[5 + 6 * f] = fr[3 * l]; r[6 + 6 * f] = fr[3 * l + 1]; r[7 + 6 * f] = fr[3 * l + 2];
                    r[8 + 6 * f] = lv[3 * l]; r[9 + 6 * f] = lv[3 * l + 1]; r[10 + 6 * f] = lv[3 * l + 2];
                }
            }
            break;
        case GF_R_FOOT_HEIGHT:
            if constexpr (HasGaitTerms<Args>::value) {   // target height, per foot: link velocity xy and link height
                const GfContactView& cv = a.contact[t.i[0]];
                const GfCommandView& gv = a.command[t.i[1]];
                const GF_GLOBAL float* lv = G(cv.link_vel) + n * cv.num_links * 3;
                const GF_GLOBAL float* lp = G(cv.link_pos) + n * cv.num_links * 3;
                r[0] = G(gv.command)[n * cmd_stride(gv) + GF_GAIT_HEIGHT];
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int l = (t.i[2] >> (8 * f)) & 0xff;
                    r[1 + 3 * f] = lv[3 * l]; r[2 + 3 * f] = lv[3 * l + 1]; r[3 + 3 * f] = lp[3 * l + 2];
                }
            }
            break;
        default: break;
    }
}

template <class Args>
__device__ __forceinline__ float term_value(const GfTerm& t, const Args& a, int64_t n, const float (&r)[kTermRowRegs]) {
    float v = 0.f;
    switch (t.op) {
        case GF_R_HAS_CONTACT:      // (mdp/rewards.py has_contact: #links over the threshold >= min_contacts)
        case GF_R_CONTACT_FORCE:    // Σ_links max(‖f‖ − threshold, 0)
        case GF_R_FEET_SLIDE: {     // Σ_links ‖v_link‖ · [‖f‖ > 1]
            const GfContactView& cv = a.contact[t.i[0]];
            const int L = cv.num_links;
            const GF_GLOBAL float* rows = G(cv.contacts) + n * L * 3;
            const GF_GLOBAL float* vels = G(cv.link_vel) + n * L * 3;
            float s = 0.f;
            int cnt = 0;
            auto four = [&](const float (&f)[4][3], const float (&w)[4][3], int l0) __attribute__((always_inline)) {   // links in order
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float fn = norm3(f[j][0], f[j][1], f[j][2]);
                    if (t.op == GF_R_HAS_CONTACT) cnt += (l0 + j < L && fn > t.p[0]) ? 1 : 0;
                    else if (t.op == GF_R_CONTACT_FORCE) { if (l0 + j < L) s += clamp_min(fn - t.p[0], 0.f); }
                    else { if (l0 + j < L) s += norm3(w[j][0], w[j][1], w[j][2]) * (fn > 1.0f ? 1.f : 0.f); }
                }
            };
            if (L > 0) {
                float f[4][3], w[4][3] = {};
#pragma unroll
                for (int j = 0; j < 4; ++j) { f[j][0] = r[3 * j]; f[j][1] = r[3 * j + 1]; f[j][2] = r[3 * j + 2]; }
                if (t.op == GF_R_FEET_SLIDE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { w[j][0] = r[12 + 3 * j]; w[j][1] = r[13 + 3 * j]; w[j][2] = r[14 + 3 * j]; }
                }
                four(f, w, 0);
            }
            for (int l0 = 4; l0 < L; l0 += 4) {
                float f[4][3], w[4][3] = {};
                link_rows4(f, rows, l0, L);
                if (t.op == GF_R_FEET_SLIDE) link_rows4(w, vels, l0, L);
                four(f, w, l0);
            }
            v = t.op == GF_R_HAS_CONTACT ? (cnt >= t.i[1] ? 1.f : 0.f) : s;
        } break;
        case GF_R_GAIT_PHASE:
            if constexpr (HasGaitTerms<Args>::value) {
                // gait_phase_reward (examples/gait_trainer/gait_command_manager.py:295-345): per foot, force is penalised in the
                // swing half of its cycle and speed in the stance half; the four foot terms add left to right (FL, FR, RL, RR).
                // The reference's index-list quirk (see gf_step.h): env 0 needs "does ANY env have foot f in swing / stance".  The gait
                // kernel keeps one such byte per 64-env block; the wave that owns env 0 ORs them (4 bytes per lane per pass + a butterfly).
                uint32_t any = 0;
                const bool quirk = a.gait_wave_flags != nullptr && blockIdx.x == 0;  // wave-uniform
                if (quirk) {
                    const int words = (int)(((int64_t)a.num_envs + 255) / 256);
                    const GF_GLOBAL uint32_t* w = reinterpret_cast<const GF_GLOBAL uint32_t*>(G(a.gait_wave_flags));
                    for (int i = (int)(threadIdx.x & (GF_WAVE - 1)); i < words; i += GF_WAVE) any |= w[i];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) any |= (uint32_t)__shfl_xor((int)any, o, GF_WAVE);
                    any = (any | (any >> 8) | (any >> 16) | (any >> 24)) & 0xffu;
                }
                const bool env0 = quirk && n == 0;
                float quad = 0.f;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const float force = norm3(r[5 + 6 * f], r[6 + 6 * f], r[7 + 6 * f]);
                    const float vel = norm3(r[8 + 6 * f], r[9 + 6 * f], r[10 + 6 * f]);
                    int fl = gait_foot_flags(r[0], r[1 + f], t.p[1], t.p[2]);  // p1 = (float)(2π), p2 = (float)π
                    if (env0) {
                        if ((any >> (2 * f + 1)) & 1u) fl = 2;
                        else if ((any >> (2 * f)) & 1u) fl = 1;
                    }
                    const float fw = (fl & 1) ? -1.0f : 0.0f, vw = (fl & 2) ? -1.0f : 0.0f;
                    const float foot = vw * vel + fw * force;
                    quad = f == 0 ? foot : quad + foot;
                }
                v = expf(quad);
            }
            break;
        case GF_R_FOOT_HEIGHT:
            if constexpr (HasGaitTerms<Args>::value) {   // foot_height_reward (:278-293): exp(-Σ_feet |v_xy| (z - foot_height)^2 / sensitivity)
                float err = 0.f;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const float d = r[3 + 3 * f] - r[0];
                    const float e = norm2(r[1 + 3 * f], r[2 + 3 * f]) * (d * d);
                    err = f == 0 ? e : err + e;
                }
                v = expf((-err) / t.p[0]);
            }
            break;
        default: break;
    }
    return v;
}

template <int OP, class Args>
__device__ __forceinline__ float term_rows_then_value(const GfTerm& t, const Args& a, int64_t n) {
    GfTerm tt = t;
    tt.op = OP;   // (== t.op: the caller switched on it)
    float rows[kTermRowRegs];
    term_rows(tt, a, n, rows);
    return term_value(tt, a, n, rows);
}

// what a reward term needs besides memory: the body-frame vectors (the quaternion rotation), the termination mask
__host__ __device__ constexpr bool reward_op_body_frame(int op) {
    return op == GF_R_LIN_VEL_Z_L2 || op == GF_R_ANG_VEL_XY_L2 || op == GF_R_FLAT_ORIENTATION_L2 || op == GF_R_BODY_ACCEL_EXP ||
           op == GF_R_CMD_TRACK_LIN_VEL || op == GF_R_CMD_TRACK_ANG_VEL;
}
__host__ __device__ constexpr bool reward_op_reads_terminated(int op) { return op == GF_R_IS_ALIVE || op == GF_R_TERMINATED; }

template <class Args>
__device__ __forceinline__ float eval_reward_term(const GfTerm& t, const Args& a, const RewardRegs& r) {
    const V3 &pos = r.pos, &blin = r.blin, &bang = r.bang, &grav = r.grav;
    const float dof_dev = r.dof_dev, act_rate = r.act_rate;
    const int terminated = r.terminated;
    const float* cmd0 = r.cmd0;
    const int64_t n = r.n;
    const bool live = r.live;
    float v = 0.f;
    switch (t.op) {
        case GF_R_IS_ALIVE: v = terminated ? 0.f : 1.f; break;
        case GF_R_TERMINATED: v = terminated ? 1.f : 0.f; break;
        case GF_R_BASE_HEIGHT: {
            float h = pos.z;
            if (t.flags & GF_RW_FLAG_TERRAIN) h = h - terrain_height(a.terrain, pos.x, pos.y);
            const float target = (t.flags & GF_RW_FLAG_CMD) ? G(a.command[t.i[0]].command)[n * cmd_stride(a.command[t.i[0]])] : t.p[0];
            const float e = h - target;
            v = e * e;
        } break;
        case GF_R_DOF_SIMILAR_TO_DEFAULT: v = dof_dev; break;
        case GF_R_LIN_VEL_Z_L2: v = blin.z * blin.z; break;
        case GF_R_ANG_VEL_XY_L2: v = bang.x * bang.x + bang.y * bang.y; break;
        case GF_R_FLAT_ORIENTATION_L2: v = grav.x * grav.x + grav.y * grav.y; break;
        case GF_R_BODY_ACCEL_EXP: {
            GF_GLOBAL float* st = G(a.state[t.i[0]]) + n * 6;
            V3 la{0, 0, 0}, aa{0, 0, 0};
            if (!(t.flags & GF_RW_FLAG_FIRST_CALL)) {
                la = V3{(blin.x - st[0]) / a.dt, (blin.y - st[1]) / a.dt, (blin.z - st[2]) / a.dt};
                aa = V3{(bang.x - st[3]) / a.dt, (bang.y - st[4]) / a.dt, (bang.z - st[5]) / a.dt};
            }
            if (live) {
                st[0] = blin.x; st[1] = blin.y; st[2] = blin.z;
                st[3] = bang.x; st[4] = bang.y; st[5] = bang.z;
            }
            const float motion = norm3(la.x, la.y, la.z) + norm3(aa.x, aa.y, aa.z);
            v = 1.0f - expf((-t.p[0]) * motion);
        } break;
        case GF_R_ACTION_RATE_L2: v = act_rate; break;
        case GF_R_CMD_TRACK_LIN_VEL: {
            float c0 = cmd0[0], c1 = cmd0[1];
            if (t.i[0] != 0) {
                const GfCommandView& c = a.command[t.i[0]];
                c0 = G(c.command)[n * cmd_stride(c)];
                c1 = G(c.command)[n * cmd_stride(c) + 1];
            }
            const float e0 = c0 - blin.x;
            const float e1 = c1 - blin.y;
            const float err = e0 * e0 + e1 * e1;
            v = expf((-err) / t.p[0]);
        } break;
        case GF_R_CMD_TRACK_ANG_VEL: {
            float cz;
            if (t.i[0] == 0 && t.i[1] < 3) {
                cz = t.i[1] == 0 ? cmd0[0] : (t.i[1] == 1 ? cmd0[1] : cmd0[2]);
            } else {
                const GfCommandView& c = a.command[t.i[0]];
                cz = G(c.command)[n * cmd_stride(c) + t.i[1]];
            }
            const float e = cz - bang.z;
            v = expf((-(e * e)) / t.p[0]);
        } break;
        case GF_R_STAND_STILL: {
            float c0 = cmd0[0], c1 = cmd0[1];
            if (t.i[0] != 0) {
                const GfCommandView& c = a.command[t.i[0]];
                c0 = G(c.command)[n * cmd_stride(c)];
                c1 = G(c.command)[n * cmd_stride(c) + 1];
            }
            const float m = norm2(c0, c1);
            v = dof_dev * ((m < t.p[0]) ? 1.f : 0.f);
        } break;
        case GF_R_FEET_AIR_TIME: {
            const GfContactView& cv = a.contact[t.i[0]];
            float s = 0.f;
            const int L = cv.num_links;
            const GF_GLOBAL float* ct = G(cv.current_contact_time) + n * L;
            const GF_GLOBAL float* at = G(cv.last_air_time) + n * L;
            for (int l0 = 0; l0 < L; l0 += 4) {   // four links per pass: loads first, then the sum in link order
                float cc[4], la[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int l = l0 + j < L ? l0 + j : L - 1;
                    cc[j] = ct[l]; la[j] = at[l];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float made = ((cc[j] > 0.f) && (cc[j] < t.p[2])) ? 1.f : 0.f;
                    float air = (la[j] - t.p[0]) * made;
                    if (t.flags & GF_RW_FLAG_MAX) air = clamp_max(air, t.p[1]);
                    if (l0 + j < L) s += air;
                }
            }
            if (t.i[1] >= 0) {
                float c0 = cmd0[0], c1 = cmd0[1];
                if (t.i[1] != 0) {
                    const GfCommandView& c = a.command[t.i[1]];
                    c0 = G(c.command)[n * cmd_stride(c)];
                    c1 = G(c.command)[n * cmd_stride(c) + 1];
                }
                s = s * ((norm2(c0, c1) > 0.1f) ? 1.f : 0.f);
            }
            v = s;
        } break;
        case GF_R_EXTERNAL: v = G(a.ext[t.i[0]])[n]; break;
        // the memory-only terms: rows, then value (see term_rows) — one case each, with the opcode a constant inside it: behind a
        // run-time opcode the two halves are two switches, and the register block would have to stay live between them for every
        // term at once (the table interpreter of the fused kernel went to scratch that way)
        case GF_R_HAS_CONTACT: v = term_rows_then_value<GF_R_HAS_CONTACT>(t, a, n); break;
        case GF_R_CONTACT_FORCE: v = term_rows_then_value<GF_R_CONTACT_FORCE>(t, a, n); break;
        case GF_R_FEET_SLIDE: v = term_rows_then_value<GF_R_FEET_SLIDE>(t, a, n); break;
        case GF_R_GAIT_PHASE: v = term_rows_then_value<GF_R_GAIT_PHASE>(t, a, n); break;
        case GF_R_FOOT_HEIGHT: v = term_rows_then_value<GF_R_FOOT_HEIGHT>(t, a, n); break;
        default: break;
    }
    return v;
}

}  // namespace gf
