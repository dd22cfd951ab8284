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
            v = contact_count_over(a.contact[t.i[0]], m, t.p[0]) >= t.i[1];
            break;
        case GF_T_CONTACT_FORCE:
            v = contact_count_over(a.contact[t.i[0]], m, t.p[0]) > 0;
            break;
        case GF_T_CONTACT_FORCE_GRACE:
            v = !(ep_len <= t.i[1]) && (contact_count_over(a.contact[t.i[0]], m, t.p[0]) > 0);
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

template <class Args>
__device__ __forceinline__ float gait_phase_term(const GfTerm& t, const Args& a, int64_t n) {
    // gait_phase_reward (examples/gait_trainer/gait_command_manager.py:295-345): per foot, force is penalised in the
    // swing half of its cycle and speed in the stance half; the four foot terms add left to right (FL, FR, RL, RR)
    const GfContactView& cv = a.contact[t.i[0]];
    const GfCommandView& gv = a.command[t.i[1]];
    const GF_GLOBAL float* g = G(gv.command) + n * cmd_stride(gv);
    const GF_GLOBAL float* fr = G(cv.contacts) + n * cv.num_links * 3;
    const GF_GLOBAL float* lv = G(cv.link_vel) + n * cv.num_links * 3;
    const float phase = g[GF_GAIT_PHASE];
    // the reference's index-list quirk (see gf_step.h): env 0 needs "does ANY env have foot f in swing / stance".  The gait
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
        const int l = (t.i[2] >> (8 * f)) & 0xff;
        const float force = norm3(fr[3 * l], fr[3 * l + 1], fr[3 * l + 2]);
        const float vel = norm3(lv[3 * l], lv[3 * l + 1], lv[3 * l + 2]);
        int fl = gait_foot_flags(phase, g[GF_GAIT_OFFSET + f], t.p[1], t.p[2]);  // p1 = (float)(2π), p2 = (float)π
        if (env0) {
            if ((any >> (2 * f + 1)) & 1u) fl = 2;
            else if ((any >> (2 * f)) & 1u) fl = 1;
        }
        const float fw = (fl & 1) ? -1.0f : 0.0f, vw = (fl & 2) ? -1.0f : 0.0f;
        const float foot = vw * vel + fw * force;
        quad = f == 0 ? foot : quad + foot;
    }
    return expf(quad);
}

template <class Args>
__device__ __forceinline__ float foot_height_term(const GfTerm& t, const Args& a, int64_t n) {
    // foot_height_reward (:278-293): exp(-Σ_feet |v_xy| (z - foot_height)^2 / sensitivity)
    const GfContactView& cv = a.contact[t.i[0]];
    const GfCommandView& gv = a.command[t.i[1]];
    const float target = G(gv.command)[n * cmd_stride(gv) + GF_GAIT_HEIGHT];
    const GF_GLOBAL float* lv = G(cv.link_vel) + n * cv.num_links * 3;
    const GF_GLOBAL float* lp = G(cv.link_pos) + n * cv.num_links * 3;
    float err = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int l = (t.i[2] >> (8 * f)) & 0xff;
        const float d = lp[3 * l + 2] - target;
        const float e = norm2(lv[3 * l], lv[3 * l + 1]) * (d * d);
        err = f == 0 ? e : err + e;
    }
    return expf((-err) / t.p[0]);
}

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
        case GF_R_HAS_CONTACT: v = contact_count_over(a.contact[t.i[0]], n, t.p[0]) >= t.i[1] ? 1.f : 0.f; break;
        case GF_R_CONTACT_FORCE: {
            const GfContactView& cv = a.contact[t.i[0]];
            const GF_GLOBAL float* r = G(cv.contacts) + n * cv.num_links * 3;
            float s = 0.f;
            for (int l = 0; l < cv.num_links; ++l) s += clamp_min(norm3(r[3 * l], r[3 * l + 1], r[3 * l + 2]) - t.p[0], 0.f);
            v = s;
        } break;
        case GF_R_FEET_AIR_TIME: {
            const GfContactView& cv = a.contact[t.i[0]];
            float s = 0.f;
            for (int l = 0; l < cv.num_links; ++l) {
                const float cc = G(cv.current_contact_time)[n * cv.num_links + l];
                const float made = ((cc > 0.f) && (cc < t.p[2])) ? 1.f : 0.f;
                float air = (G(cv.last_air_time)[n * cv.num_links + l] - t.p[0]) * made;
                if (t.flags & GF_RW_FLAG_MAX) air = clamp_max(air, t.p[1]);
                s += air;
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
        case GF_R_FEET_SLIDE: {
            const GfContactView& cv = a.contact[t.i[0]];
            const GF_GLOBAL float* r = G(cv.contacts) + n * cv.num_links * 3;
            const GF_GLOBAL float* lv = G(cv.link_vel) + n * cv.num_links * 3;
            float s = 0.f;
            for (int l = 0; l < cv.num_links; ++l) {
                const float c = norm3(r[3 * l], r[3 * l + 1], r[3 * l + 2]) > 1.0f ? 1.f : 0.f;
                s += norm3(lv[3 * l], lv[3 * l + 1], lv[3 * l + 2]) * c;
            }
            v = s;
        } break;
        case GF_R_EXTERNAL: v = G(a.ext[t.i[0]])[n]; break;
        case GF_R_GAIT_PHASE:
            if constexpr (HasGaitTerms<Args>::value) v = gait_phase_term(t, a, n);
            break;
        case GF_R_FOOT_HEIGHT:
            if constexpr (HasGaitTerms<Args>::value) v = foot_height_term(t, a, n);
            break;
        default: break;
    }
    return v;
}

}  // namespace gf
