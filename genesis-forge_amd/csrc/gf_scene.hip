// gf_scene.hip — EntityManager body-frame getters as a standalone call, and the synthetic scene tick.
//
// gf_entity_rotate: EntityManager.get_projected_gravity / get_linear_velocity / get_angular_velocity
//   (managers/entity_manager.py:130-146, utils.py:13-55) for callers that ask for one vector outside a
//   fused phase (opaque user terms).  7-10 torch launches per call in the reference.
//
// gf_synth_scene_step: stands in for Genesis' scene.step() (managed_env.py:292) so the manager
//   pipeline can be driven, benchmarked and parity-tested without the simulator.
//   It is NOT physics: joints track their PD targets with a first-order lag, the base does a damped
//   random walk driven by Philox draws, contacts are sampled per slot.  What matters is that the
//   update is deterministic and integer-RNG driven with f32 ops in a fixed order, so this kernel, the
//   C oracle and the numpy model that drives the reference produce the same bits.
#include "gf_launch.h"

namespace gf {

__global__ __launch_bounds__(kEnvBlock) void rotate_kernel(const GfRotateArgs a) {
    const int64_t n = (int64_t)blockIdx.x * kEnvBlock + threadIdx.x;
    if (n >= a.num_envs) return;
    const float4 q = load_quat(a.entity.quat, n);
    V3 v{0.f, 0.f, -1.f};
    if (a.what == GF_ROT_LIN_VEL) v = load3(a.entity.lin_vel, n);
    else if (a.what == GF_ROT_ANG_VEL) v = load3(a.entity.ang_vel, n);
    const V3 o = rot_inv(q, v);
    float* r = a.out + 3 * n;
    r[0] = o.x; r[1] = o.y; r[2] = o.z;
}

// Where the synthetic scene puts link l relative to the base: scene link 0 is the ground entity, 1 the robot's base,
// then chains of four (hip, thigh, calf, foot) hanging 8.5 cm apart below the four corners of the body.
struct LinkOffset { float x, y, z; };
__device__ __forceinline__ LinkOffset synth_link_offset(int l) {
    if (l <= 1) return {0.0f, 0.0f, 0.0f};
    const int leg = (l - 2) / 4, depth = (l - 2) % 4 + 1;
    return {leg < 2 ? 0.19f : -0.19f, (leg & 1) ? -0.11f : 0.11f, -0.085f * (float)depth};
}

// the base state an env's tick leaves: position, quaternion, linear and angular velocity
struct SynthBase {
    float p[3], q[4], v[3], w[3];
};

template <int DV>
__device__ __forceinline__ void synth_state_body(const GfSynthSceneArgs& a, const int64_t n, SynthBase& out) {
    const int D = a.num_dofs;
    const float dt = a.dt;
    const uint32_t genv = (uint32_t)n + a.env_offset;
    // all loads first: joint rows as float4 (DV = D/4 when the rows are 16-byte tiles), base state, then one wait
    float4 tg[DV > 0 ? DV : 1], dp[DV > 0 ? DV : 1];
    if (DV > 0) {
        const float4* t4 = reinterpret_cast<const float4*>(a.targets + n * D);
        const float4* p4 = reinterpret_cast<const float4*>(a.dof_pos + n * D);
#pragma unroll
        for (int c = 0; c < DV; ++c) { tg[c] = t4[c]; dp[c] = p4[c]; }
    }
    const float4 q4 = load_quat(a.quat, n);
    const V3 w0 = load3(a.ang_vel, n), v0 = load3(a.lin_vel, n), p0 = load3(a.pos, n);
    // six base-motion draws = Philox columns 0..5 = block 0 (x,y,z,w) + block 1 (x,y)
    const U4 b0 = philox4x32_10(genv, 0u, (uint32_t)a.tick, (uint32_t)(a.tick >> 32), (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const U4 b1 = philox4x32_10(genv, 1u, (uint32_t)a.tick, (uint32_t)(a.tick >> 32), (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const float s[6] = {u24_to_unit(b0.x) * 2.0f - 1.0f, u24_to_unit(b0.y) * 2.0f - 1.0f, u24_to_unit(b0.z) * 2.0f - 1.0f,
                        u24_to_unit(b0.w) * 2.0f - 1.0f, u24_to_unit(b1.x) * 2.0f - 1.0f, u24_to_unit(b1.y) * 2.0f - 1.0f};
    if (DV > 0) {
        float4* v4 = reinterpret_cast<float4*>(a.dof_vel + n * D);
        float4* p4 = reinterpret_cast<float4*>(a.dof_pos + n * D);
#pragma unroll
        for (int c = 0; c < DV; ++c) {
            float4 v, p;
            v.x = (tg[c].x - dp[c].x) * a.joint_rate; p.x = dp[c].x + v.x * dt;
            v.y = (tg[c].y - dp[c].y) * a.joint_rate; p.y = dp[c].y + v.y * dt;
            v.z = (tg[c].z - dp[c].z) * a.joint_rate; p.z = dp[c].z + v.z * dt;
            v.w = (tg[c].w - dp[c].w) * a.joint_rate; p.w = dp[c].w + v.w * dt;
            v4[c] = v;
            p4[c] = p;
        }
    } else {
        for (int d = 0; d < D; ++d) {
            const float cur = a.dof_pos[n * D + d];
            const float err = a.targets[n * D + d] - cur;
            const float v = err * a.joint_rate;
            a.dof_vel[n * D + d] = v;
            a.dof_pos[n * D + d] = cur + v * dt;
        }
    }
    float* wp = a.ang_vel + 3 * n;
    float* vp = a.lin_vel + 3 * n;
    float* pp = a.pos + 3 * n;
    float* qp = a.quat + 4 * n;
    float w[3] = {w0.x, w0.y, w0.z}, v[3] = {v0.x, v0.y, v0.z}, p[3] = {p0.x, p0.y, p0.z};
#pragma unroll
    for (int j = 0; j < 3; ++j) w[j] = w[j] * 0.9f + s[j] * a.ang_noise;
    v[0] = v[0] * 0.9f + s[3] * a.lin_noise;
    v[1] = v[1] * 0.9f + s[4] * a.lin_noise;
    v[2] = (v[2] * 0.9f + (a.height_target - p[2]) * 2.0f) + s[5] * a.lin_noise;
#pragma unroll
    for (int j = 0; j < 3; ++j) p[j] = p[j] + v[j] * dt;
    const float h = 0.5f * dt;
    const float qw = q4.x, qx = q4.y, qy = q4.z, qz = q4.w;
    const float dw = ((-(w[0] * qx)) - w[1] * qy) - w[2] * qz;
    const float dx = (w[0] * qw + w[1] * qz) - w[2] * qy;
    const float dy = (w[1] * qw + w[2] * qx) - w[0] * qz;
    const float dz = (w[2] * qw + w[0] * qy) - w[1] * qx;
    float nq[4] = {qw + dw * h, qx + dx * h, qy + dy * h, qz + dz * h};
    const float nrm = sqrtf(((nq[0] * nq[0] + nq[1] * nq[1]) + nq[2] * nq[2]) + nq[3] * nq[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) nq[j] = nq[j] / nrm;
#pragma unroll
    for (int j = 0; j < 3; ++j) { wp[j] = w[j]; vp[j] = v[j]; pp[j] = p[j]; out.w[j] = w[j]; out.v[j] = v[j]; out.p[j] = p[j]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { qp[j] = nq[j]; out.q[j] = nq[j]; }
}

template <int DV>
__global__ __launch_bounds__(kEnvBlock) void synth_scene_kernel(const GfSynthSceneArgs a) {
    const int64_t n = (int64_t)blockIdx.x * kEnvBlock + threadIdx.x;
    if (n >= a.num_envs) return;
    SynthBase b;
    synth_state_body<DV>(a, n, b);
}

// Per-link outputs (orientation, velocity, position of scene link l of env n; flat index gid = n·NL + l) from the base state the
// tick just produced: consecutive lanes write consecutive floats.
__device__ __forceinline__ void synth_link_one(const GfSynthSceneArgs& a, const int64_t gid, const int l, const float* b /* p3 q4 v3 w3 */) {
    const V3 p{b[0], b[1], b[2]}, v{b[7], b[8], b[9]}, w{b[10], b[11], b[12]};
    if (a.links_quat_out) reinterpret_cast<float4*>(a.links_quat_out)[gid] = make_float4(b[3], b[4], b[5], b[6]);
    if (a.links_vel_out) {
        float* o = a.links_vel_out + gid * 3;
        o[0] = v.x + (float)(l % 3 == 0 ? 1 : 0) * 0.05f * w.x;
        o[1] = v.y + (float)(l % 3 == 1 ? 1 : 0) * 0.05f * w.y;
        o[2] = v.z + (float)(l % 3 == 2 ? 1 : 0) * 0.05f * w.z;
    }
    if (a.links_pos_out) {
        const LinkOffset o = synth_link_offset(l);
        const float wl = l % 3 == 0 ? w.x : (l % 3 == 1 ? w.y : w.z);
        float* lp = a.links_pos_out + gid * 3;
        lp[0] = p.x + o.x;
        lp[1] = p.y + o.y;
        lp[2] = (p.z + o.z) + 0.03f * wl;
    }
}

// Sampled contact of slot c of env n (flat index k = n·C + c): 32 B of output per lane, coalesced.  The second Philox block only
// feeds the z force of an ACTIVE slot; an empty slot's outputs are constants, so it is skipped for waves without an active slot.
// foot_link_mask != 0 selects the walking model (gf_step.h, GfSynthSceneArgs): the k-th foot owns slot k and touches the ground in
// the stance half of a 20-tick trot cycle; the ground sits on side a or side b of a contact at random.
__device__ __forceinline__ void synth_contact_one(const GfSynthSceneArgs& a, const int64_t k, const uint32_t genv, const int c, const float p0, const float p1) {
    const int NL = a.num_scene_links;
    const uint32_t col = (uint32_t)(8 + 8 * c);
    // columns col..col+3 share one Philox block, col+4 starts the next
    const U4 r0 = philox4x32_10(genv, col >> 2, (uint32_t)a.tick, (uint32_t)(a.tick >> 32), (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    const float u_act = u24_to_unit(r0.x), u_link = u24_to_unit(r0.y);
    const uint32_t feet = a.foot_link_mask;
    bool active = u_act < a.contact_prob;
    int lb = 1 + (int)(u_link * (float)(NL - 1));
    bool robot_on_a = false, foot_slot = false;
    float fz = 0.0f;
    if (feet) {
        const int n_feet = __builtin_popcount(feet);
        if (c < n_feet) {
            uint32_t m = feet;
            for (int j = 0; j < c; ++j) m &= m - 1u;
            lb = __builtin_ctz(m);
            const uint32_t ph = ((uint32_t)a.tick + genv * 7u) % 20u;
            const bool pair_a = ((c ^ (c >> 1)) & 1) == 0;
            const bool stance = pair_a ? ph < 10u : ph >= 10u;
            const float p = stance ? fminf(1.8f * a.foot_contact_prob, 1.0f) : 0.2f * a.foot_contact_prob;
            active = u_act < p;
            robot_on_a = u_link < 0.5f;
            // a foot slot takes its normal force from the same draw as its side (both halves of [0, 1) map onto [0, 1), exactly):
            // every wave holds foot slots, so a second Philox block for them would be paid by every lane of every wave
            fz = robot_on_a ? u_link * 2.0f : u_link * 2.0f - 1.0f;
            foot_slot = true;
        } else {
            const int kk = (int)(u_link * (float)(2 * (NL - 1)));
            lb = 1 + (kk >> 1);
            robot_on_a = (kk & 1) != 0;
        }
    }
    if (active && !foot_slot) {
        const U4 r1 = philox4x32_10(genv, (col >> 2) + 1, (uint32_t)a.tick, (uint32_t)(a.tick >> 32), (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
        fz = u24_to_unit(r1.x);
    }
    const float fx = u24_to_unit(r0.z) * 2.0f - 1.0f, fy = u24_to_unit(r0.w) * 2.0f - 1.0f;
    if (lb > NL - 1) lb = NL - 1;
    // the stored force is the force on link_b: with the robot link on side a it is the reaction
    const float sx = fx * a.contact_force * 0.25f, sy = fy * a.contact_force * 0.25f, sz = fz * a.contact_force;
    a.link_a_out[k] = active ? (robot_on_a ? lb : 0) : -1;
    a.link_b_out[k] = active ? (robot_on_a ? 0 : lb) : -1;
    a.contact_force_out[k * 3 + 0] = active ? (robot_on_a ? -sx : sx) : 0.0f;
    a.contact_force_out[k * 3 + 1] = active ? (robot_on_a ? -sy : sy) : 0.0f;
    a.contact_force_out[k * 3 + 2] = active ? (robot_on_a ? -sz : sz) : 0.0f;
    a.contact_pos_out[k * 3 + 0] = active ? p0 + fx * 0.2f : 0.0f;
    a.contact_pos_out[k * 3 + 1] = active ? p1 + fy * 0.2f : 0.0f;
    a.contact_pos_out[k * 3 + 2] = 0.0f;
}

// A scene with per-link outputs and / or contacts, as ONE launch (two until round 2: the state tick, then a flat kernel over
// (env, link) and (env, slot) pairs that read the new base state back): workgroup b owns envs [64b, 64b+64) — wave 0 ticks them
// (lane = env) and leaves their new base state in LDS, then all four waves write the tile's link rows and contact slots, which are
// contiguous in memory.  Same statements per element, so the outputs are bit-identical to the two-launch version and the oracle.
// TE = envs per workgroup: 64 (the tile of the action / post-physics kernels, same XCD) when that still fills the chip, 16 below
// ~32 k envs (4 096 envs are 64 tiles of 64 — a quarter of the CUs — but 256 tiles of 16).
constexpr int kSynthTileBlock = 256;
template <int DV, int TE>
__global__ __launch_bounds__(kSynthTileBlock) void synth_tile_kernel(const GfSynthSceneArgs a, const int links, const int contacts) {
    __shared__ float s_base[TE][13];   // p(3) q(4) v(3) w(3)
    const int64_t n0 = (int64_t)blockIdx.x * TE;
    const int rows = (int)((int64_t)a.num_envs - n0 < TE ? (int64_t)a.num_envs - n0 : TE);
    const int tid = threadIdx.x;
    if (tid < rows) {
        SynthBase b;
        synth_state_body<DV>(a, n0 + tid, b);
        float* r = s_base[tid];
        r[0] = b.p[0]; r[1] = b.p[1]; r[2] = b.p[2];
        r[3] = b.q[0]; r[4] = b.q[1]; r[5] = b.q[2]; r[6] = b.q[3];
        r[7] = b.v[0]; r[8] = b.v[1]; r[9] = b.v[2];
        r[10] = b.w[0]; r[11] = b.w[1]; r[12] = b.w[2];
    }
    __syncthreads();
    if (links) {
        const int NL = a.num_scene_links;
        for (int i = tid; i < rows * NL; i += kSynthTileBlock) {
            const int e = i / NL, l = i - e * NL;
            synth_link_one(a, n0 * NL + i, l, s_base[e]);
        }
    }
    if (contacts) {
        const int C = a.num_contacts;
        for (int i = tid; i < rows * C; i += kSynthTileBlock) {
            const int e = i / C, c = i - e * C;
            synth_contact_one(a, n0 * C + i, (uint32_t)(n0 + e) + a.env_offset, c, s_base[e][0], s_base[e][1]);
        }
    }
}

}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_entity_rotate(const GfRotateArgs* a, void* stream) {
    if (!a || !a->out || !a->entity.quat) return GF_E_NULL;
    if (a->what < GF_ROT_PROJ_GRAVITY || a->what > GF_ROT_ANG_VEL || a->num_envs < 0) return GF_E_RANGE;
    if (a->what == GF_ROT_LIN_VEL && !a->entity.lin_vel) return GF_E_NULL;
    if (a->what == GF_ROT_ANG_VEL && !a->entity.ang_vel) return GF_E_NULL;
    if (reinterpret_cast<uintptr_t>(a->entity.quat) & 15u) return GF_E_UNSUPPORTED;
    if (a->num_envs == 0) return GF_OK;
    hipStream_t s = (hipStream_t)stream;
    gf::PhaseScope scope(GF_PHASE_ROTATE, s);
    scope.begin_bracket();
    gf::klaunch(gf::rotate_kernel, dim3(gf::env_grid(a->num_envs)), dim3(gf::kEnvBlock), 0, s, *a);
    return gf::launch_status();
}

extern "C" __attribute__((visibility("default"))) int gf_synth_scene_step(const GfSynthSceneArgs* a, void* stream) {
    if (!a || !a->pos || !a->quat || !a->lin_vel || !a->ang_vel || !a->dof_pos || !a->dof_vel || !a->targets) return GF_E_NULL;
    if (a->num_envs < 0 || a->num_dofs <= 0 || a->num_contacts < 0) return GF_E_RANGE;
    if (a->num_contacts > 0 && a->contact_force_out && (!a->contact_pos_out || !a->link_a_out || !a->link_b_out || a->num_scene_links < 2)) return GF_E_NULL;
    if (reinterpret_cast<uintptr_t>(a->quat) & 15u) return GF_E_UNSUPPORTED;
    if (a->links_quat_out && (reinterpret_cast<uintptr_t>(a->links_quat_out) & 15u)) return GF_E_UNSUPPORTED;
    if (a->num_envs == 0) return GF_OK;
    hipStream_t s = (hipStream_t)stream;
    gf::PhaseScope scope(GF_PHASE_SCENE, s);
    scope.begin_bracket();
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const bool rows16 = (a->num_dofs % 4 == 0) && al16(a->targets) && al16(a->dof_pos) && al16(a->dof_vel);
    const unsigned grid = gf::env_grid(a->num_envs);
    const bool links = (a->links_quat_out || a->links_vel_out || a->links_pos_out) && a->num_scene_links > 0;
    const bool contacts = a->num_contacts > 0 && a->contact_force_out;
    if (links || contacts) {   // tick + per-link rows + contact slots of a 64-env tile in one launch
        const dim3 tb(gf::kSynthTileBlock);
        const bool small = a->num_envs < 32768;
        const dim3 tg(small ? gf::env_grid(a->num_envs, 16) : grid);
        const int dv = !rows16 ? 0 : (a->num_dofs == 12 ? 3 : (a->num_dofs == 28 ? 7 : 0));
#define GF_SYNTH_TILE(DVV)                                                                                              \
        if (small) gf::klaunch(gf::synth_tile_kernel<DVV, 16>, tg, tb, 0, s, *a, (int)links, (int)contacts);          \
        else gf::klaunch(gf::synth_tile_kernel<DVV, 64>, tg, tb, 0, s, *a, (int)links, (int)contacts)
        if (dv == 3) { GF_SYNTH_TILE(3); } else if (dv == 7) { GF_SYNTH_TILE(7); } else { GF_SYNTH_TILE(0); }
#undef GF_SYNTH_TILE
    } else if (rows16 && a->num_dofs == 12) gf::klaunch(gf::synth_scene_kernel<3>, dim3(grid), dim3(gf::kEnvBlock), 0, s, *a);
    else if (rows16 && a->num_dofs == 28) gf::klaunch(gf::synth_scene_kernel<7>, dim3(grid), dim3(gf::kEnvBlock), 0, s, *a);
    else gf::klaunch(gf::synth_scene_kernel<0>, dim3(grid), dim3(gf::kEnvBlock), 0, s, *a);
    return gf::launch_status();
}
