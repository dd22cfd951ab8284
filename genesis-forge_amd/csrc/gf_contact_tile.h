// gf_contact_tile.h — ContactManager.step for the envs of one workgroup: the body shared by the stand-alone contact launch
// (gf_contact.hip: E envs per workgroup, sized for parallelism) and the fused post-physics launch (gf_post_ws.h: the 64-env tile of
// the step's other phases — the contact phase runs in front of them in the SAME launch, SURVEY.md §8f-1 ∘ §8f-2).
//
// Replaces, per ContactManager (the gait config has three):
//   managers/contact/contact_manager.py:399-403   isnan/isinf .any() (2 host syncs) + nan_to_num
//   managers/contact/contact_manager.py:408-411   3 fill_ launches
//   managers/contact/kernel.py:35-90              the Taichi kernel (atomic += over (env, contact, target))
//   managers/contact/contact_manager.py:434-477   norm > threshold + 4 torch.where air-time updates
//
#pragma once

#include "gf_device.h"

namespace gf {

#define GF_CONTACT_INLINE __attribute__((always_inline))
typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));   // dword aligned; one global_{load,store}_dwordx3: a wave moves 768 contiguous bytes
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st3(float* p, float x, float y, float z) { *reinterpret_cast<GF_GLOBAL f32x3*>(G(p)) = f32x3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const float* p) {
    const f32x3 v = *reinterpret_cast<const GF_GLOBAL f32x3*>(G(p));
    return V3{v.x, v.y, v.z};
}

// One manager as the lanes read it (an LDS image: a lane's manager is lane-dependent, and picking 14 fields x 4 managers with
// constant-index select chains out of SGPRs cost ~250 of the kernel's ~1 100 instructions per wave and 127 spilled SGPRs — on a
// kernel PMC counters show to be issue-bound, profiles/r02_i_pmc_characterization.md)
struct ContactMgrL {
    int32_t num_targets, num_with, has_with_filter, track_air_time;
    float air_time_threshold;
    int32_t _pad;
    float *contacts, *contact_positions, *position_counts, *link_vel_out, *link_pos_out;
    float *last_air_time, *current_air_time, *last_contact_time, *current_contact_time;
};
constexpr int kContactMgrWords = (int)(sizeof(ContactMgrL) / 4);
constexpr int kContactMaxMgr = 4;
constexpr int kContactMaxTargets = 64;   // tracked links over all managers of one launch

// the scene's contact arrays (collider.get_contacts(), rigid_solver.get_links_quat(), contact_manager.py:384-400)
struct ContactScene {
    const float *force, *position, *links_quat, *links_vel, *links_pos;
    const int32_t *link_a, *link_b;
    int32_t C, NL, T;   // contact slots per env, links of the scene, tracked links over all managers
    float dt;
};
// the workgroup's LDS: slot ids + occupancy masks (filled here) and the tables the caller staged
struct ContactLds {
    uint32_t* ids;               // [E*C] link_a | link_b << 16 (0xffff: no link on that side)
    uint32_t* smask;             // [E][ceil(C / 32)] slots that hold a contact
    const ContactMgrL* mgr;      // [num managers]
    const int32_t* target;       // [T] scene link id of tracked link t
    const uint16_t* meta;        // [T] manager | local index << 8
    const int32_t* with;         // [num managers][GF_MAX_LINK_IDS]
};
__host__ __device__ constexpr int contact_lds_ints(int E, int C) { return E * (C + (C + 31) / 32); }
__device__ __forceinline__ ContactLds contact_lds_carve(int32_t* p, int E, int C, const ContactMgrL* mgr, const int32_t* target, const uint16_t* meta, const int32_t* with) {
    ContactLds l;
    l.ids = reinterpret_cast<uint32_t*>(p);
    l.smask = l.ids + E * C;
    l.mgr = mgr; l.target = target; l.meta = meta; l.with = with;
    return l;
}

// floor(i / d) by multiply-shift, exact for i < 2^40 / d (i < 64·C here)
struct FastDivC {
    uint64_t m;
    uint32_t d;
    __device__ __forceinline__ explicit FastDivC(int div) : m(div > 1 ? ((1ull << 40) + (uint64_t)div - 1ull) / (uint64_t)div : 0ull), d((uint32_t)div) {}
    __device__ __forceinline__ int div(int i) const { return d > 1 ? (int)(((uint64_t)(uint32_t)i * m) >> 40) : i; }
};

struct ContactNoStamp { __device__ __forceinline__ void operator()(int) const {} };

// The first U units (four slot ids per lane and array) of a workgroup's slot-id rows, REQUESTED: the caller asks for them before it
// stages its tables (those come from the kernel-argument segment with vector loads of their own), so the two round trips are one.
template <int U>
struct ContactIdsPending {
    i32x4 va[U], vb[U];
    bool vec;
};
template <int U>
__device__ __forceinline__ ContactIdsPending<U> contact_ids_request(const ContactScene& a, const int64_t n0, const int envs_here, const int tid, const int nthreads) {
    ContactIdsPending<U> p;
    const int C = a.C;
    const GF_GLOBAL int32_t* ga = G(a.link_a) + n0 * C;
    const GF_GLOBAL int32_t* gb = G(a.link_b) + n0 * C;
    // four ids per lane and array (dwordx4: 1 KiB per wave instruction) when the block's rows start 16-byte aligned
    p.vec = C > 0 && ((n0 * C) & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.link_a) | reinterpret_cast<uintptr_t>(a.link_b)) & 15u) == 0;
    const int slots4 = p.vec ? ((envs_here * C) >> 2) : 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i4 = u * nthreads + tid;
        p.va[u] = i32x4{-1, -1, -1, -1}; p.vb[u] = p.va[u];
        if (i4 < slots4) { p.va[u] = reinterpret_cast<const GF_GLOBAL i32x4*>(ga)[i4]; p.vb[u] = reinterpret_cast<const GF_GLOBAL i32x4*>(gb)[i4]; }
    }
    return p;
}

// ContactManager.step for the E (<= 64, <= nthreads) envs [n0, n0 + envs_here) of a workgroup.
// (1) Their link_a / link_b rows (E·C ints each, contiguous in memory) are staged into LDS as packed pairs with flat coalesced
// loads — every slot id is read from HBM exactly once, whatever the number of managers and tracked links, all of a lane's requests
// in flight together — and every slot that holds a contact (ids >= 0; contacts are sparse) sets its bit in the env's occupancy
// mask (an LDS OR: order independent).  (2) One lane per (env, tracked link) walks ONLY the occupied slots of its env, in slot order
// (ctz over the mask), accumulators in registers: no float atomics, a fixed f32 sum order (the Taichi reference's atomic order is
// not).  Force, position and the target link's quaternion are fetched for matching slots only (a matching slot always involves the
// target link itself: the only quaternion that can be needed, kernel.py:74-78); the norm / threshold / air-time update the reference
// runs afterwards as ~12 launches is done by the same lane while the summed force is in registers.
// Lanes are LINK-major within a pass: nthreads / E links of every env per pass (the fused launch's 64-env tile: four — the feet
// of a quadruped are one pass, so the matches of a walking robot cost ONE dependent round trip per tile, and the passes over body
// links, which rarely touch anything, only scan LDS and store); the air-time state of the first kAirAhead passes is requested up
// front, a lane's slot-id requests go out U units at a time (the stand-alone launch sizes its workgroups for ONE pass and one unit).
// Every thread of the workgroup calls this (it synchronises the workgroup twice); the caller's tables must be written before the
// call (its first barrier publishes them).  Returns a bit per manager: this lane sanitised a non-finite force for it
// (contact_manager.py:399-403 prints a warning).
template <int kAirAhead = 4, int U = 4, class Stamp = ContactNoStamp>
__device__ __forceinline__ int contact_tile(const ContactScene& a, const ContactLds& l, const int E, const int64_t n0, const int envs_here, const int tid,
                                            const int nthreads, const ContactIdsPending<U>& pend, Stamp&& stamp = Stamp()) {
    const int C = a.C, T = a.T;
    const int MW = (C + 31) >> 5;  // 32-bit words of one env's "slot holds a contact" mask
    const int slots = envs_here * C;
    uint32_t* const ids = l.ids;
    uint32_t* const smask = l.smask;
    const int LPP = nthreads / E;                     // tracked links of every env per pass (E <= 64 <= nthreads)
    const int passes = (T + LPP - 1) / LPP;
    const FastDivC dL(LPP);
    const int e = dL.div(tid), j = tid - e * LPP;     // this lane's env and its link within a pass
    const bool lane_on = e < envs_here && e < E;
    const int64_t n = n0 + (lane_on ? e : 0);
    for (int i = tid; i < envs_here * MW; i += nthreads) smask[i] = 0u;
    __syncthreads();
    // ---- requests that depend on nothing: the air-time state of the first kAirAhead passes ---------------------------------------------
    float air_pre[kAirAhead], con_pre[kAirAhead];
#pragma unroll
    for (int p = 0; p < kAirAhead; ++p) {
        air_pre[p] = 0.f; con_pre[p] = 0.f;
        const int t = p * LPP + j;
        if (lane_on && t < T) {
            const int meta = l.meta[t];
            const ContactMgrL& m = l.mgr[meta & 0xff];
            if (m.track_air_time) {
                const int64_t k = n * m.num_targets + (meta >> 8);
                air_pre[p] = G(m.current_air_time)[k];
                con_pre[p] = G(m.current_contact_time)[k];
            }
        }
    }
    {   // ---- (1) the E rows of slot ids are contiguous — flat coalesced copy; slots that hold a contact set their bit
        const GF_GLOBAL int32_t* ga = G(a.link_a) + n0 * C;
        const GF_GLOBAL int32_t* gb = G(a.link_b) + n0 * C;
        const FastDivC dc(C);
        auto mark = [&](int i, int la, int lb) GF_CONTACT_INLINE {
            if (la >= 0 || lb >= 0) {
                const int ee = dc.div(i), c = i - ee * C;
                atomicOr(&smask[ee * MW + (c >> 5)], 1u << (c & 31));
            }
        };
        auto pack = [](int la, int lb) GF_CONTACT_INLINE { return ((uint32_t)la & 0xffffu) | ((uint32_t)lb << 16); };
        // the first U units per lane were requested by the caller (contact_ids_request) before it staged its tables; any further
        // ones — more than U·nthreads units: beyond a 64-env tile of 64 slots — are requested here, U at a time
        const bool vec = pend.vec;
        const int slots4 = vec ? (slots >> 2) : 0;
        for (int b4 = 0; b4 < slots4; b4 += U * nthreads) {
            i32x4 va[U], vb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i4 = b4 + u * nthreads + tid;
                va[u] = pend.va[u]; vb[u] = pend.vb[u];
                if (b4 > 0) {
                    va[u] = i32x4{-1, -1, -1, -1}; vb[u] = va[u];
                    if (i4 < slots4) { va[u] = reinterpret_cast<const GF_GLOBAL i32x4*>(ga)[i4]; vb[u] = reinterpret_cast<const GF_GLOBAL i32x4*>(gb)[i4]; }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i4 = b4 + u * nthreads + tid;
                if (i4 >= slots4) continue;
                reinterpret_cast<i32x4*>(ids)[i4] = i32x4{(int)pack(va[u].x, vb[u].x), (int)pack(va[u].y, vb[u].y), (int)pack(va[u].z, vb[u].z), (int)pack(va[u].w, vb[u].w)};
                // contacts are sparse: one (rarely taken) branch per four slots instead of four
                const int occ4 = ((va[u].x & vb[u].x) >= 0 ? 1 : 0) | ((va[u].y & vb[u].y) >= 0 ? 2 : 0) | ((va[u].z & vb[u].z) >= 0 ? 4 : 0) | ((va[u].w & vb[u].w) >= 0 ? 8 : 0);
                if (occ4) {
                    const int i = i4 << 2;
                    if (occ4 & 1) mark(i, va[u].x, vb[u].x);
                    if (occ4 & 2) mark(i + 1, va[u].y, vb[u].y);
                    if (occ4 & 4) mark(i + 2, va[u].z, vb[u].z);
                    if (occ4 & 8) mark(i + 3, va[u].w, vb[u].w);
                }
            }
        }
        for (int i = (slots4 << 2) + tid; i < slots; i += nthreads) {
            const int la = ga[i], lb = gb[i];
            ids[i] = pack(la, lb);
            mark(i, la, lb);
        }
    }
    __syncthreads();
    stamp(0);
    // ---- (2) one lane per (env, tracked link): ONLY the occupied slots of the env, in slot order --------------------------------------------
    int flag_mask = 0;  // bit m: this lane sanitised a non-finite force for manager m
    for (int p = 0; p < passes; ++p) {
        const int t = p * LPP + j;
        if (!lane_on || t >= T) continue;
        const int target = l.target[t];
        const int meta = l.meta[t];
        const int mi = meta & 0xff, lt = meta >> 8;
        const ContactMgrL& mg = l.mgr[mi];
        const int L = mg.num_targets, W = mg.num_with;
        const int64_t k = n * L + lt;
        float cur_air = 0.f, cur_con = 0.f;
        if (p < kAirAhead) {
#pragma unroll
            for (int pp = 0; pp < kAirAhead; ++pp)
                if (pp == p) { cur_air = air_pre[pp]; cur_con = con_pre[pp]; }
        } else if (mg.track_air_time) {
            cur_air = G(mg.current_air_time)[k]; cur_con = G(mg.current_contact_time)[k];
        }
        // the compact per-manager copies of the tracked link's velocity (feet_slide) and position (the gait manager's foot_height_reward)
        V3 lvel{0.f, 0.f, 0.f}, lpos{0.f, 0.f, 0.f};
        const bool copy_vel = a.links_vel && mg.link_vel_out, copy_pos = a.links_pos && mg.link_pos_out;
        if (copy_vel) lvel = ld3(a.links_vel + 3 * (n * a.NL + target));
        if (copy_pos) lpos = ld3(a.links_pos + 3 * (n * a.NL + target));
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, cnt = 0.f;
        const uint32_t* id_row = ids + e * C;
        for (int wd = 0; wd < MW; ++wd) {
            uint32_t bits = smask[e * MW + wd];
            while (bits) {
                const int c = (wd << 5) + __builtin_ctz(bits);
                bits &= bits - 1u;
                const uint32_t pk = id_row[c];
                const int la = (int)(int16_t)(pk & 0xffffu), lb = (int)(int16_t)(pk >> 16);
                const bool is_a = la == target, is_b = lb == target;
                if (!(is_a || is_b)) continue;
                bool include = true;
                if (mg.has_with_filter) {
                    include = false;
                    for (int w = 0; w < W; ++w) {
                        const int wl = l.with[mi * GF_MAX_LINK_IDS + w];
                        if ((is_a && lb == wl) || (is_b && la == wl)) { include = true; break; }
                    }
                }
                if (!include) continue;
                const V3 fv = ld3(a.force + (n * C + c) * 3), pv = ld3(a.position + (n * C + c) * 3);
                const float4 q = ldg4(G(a.links_quat) + (n * a.NL + target) * 4);
                float fx = fv.x, fy = fv.y, fz = fv.z;
                // torch.nan_to_num(force, nan=0, posinf=0, neginf=0)   contact_manager.py:401-403
                if (isnan(fx) || isinf(fx)) { fx = 0.f; flag_mask |= 1 << mi; }
                if (isnan(fy) || isinf(fy)) { fy = 0.f; flag_mask |= 1 << mi; }
                if (isnan(fz) || isinf(fz)) { fz = 0.f; flag_mask |= 1 << mi; }
                p0 += pv.x; p1 += pv.y; p2 += pv.z;
                cnt += 1.0f;
                // force is expressed on link_b; on link_a it is the reaction (kernel.py:74-78)
                const V3 r = is_b ? rot_inv(q, V3{fx, fy, fz}) : rot_inv(q, V3{-fx, -fy, -fz});
                f0 += r.x; f1 += r.y; f2 += r.z;
            }
        }
        st3(mg.contacts + 3 * k, f0, f1, f2);
        if (mg.contact_positions)  // kernel.py:84-90
            st3(mg.contact_positions + 3 * k, cnt > 0.f ? p0 / cnt : p0, cnt > 0.f ? p1 / cnt : p1, cnt > 0.f ? p2 / cnt : p2);
        if (mg.position_counts) G(mg.position_counts)[k] = cnt;
        if (copy_vel) st3(mg.link_vel_out + 3 * k, lvel.x, lvel.y, lvel.z);
        if (copy_pos) st3(mg.link_pos_out + 3 * k, lpos.x, lpos.y, lpos.z);
        if (mg.track_air_time) {  // contact_manager.py:441-477
            const float dt = a.dt;
            const bool is_contact = norm3(f0, f1, f2) > mg.air_time_threshold;
            const bool new_contact = (cur_air > 0.f) && is_contact;
            const bool new_detach = (cur_con > 0.f) && !is_contact;
            if (new_contact) G(mg.last_air_time)[k] = cur_air + dt;
            G(mg.current_air_time)[k] = !is_contact ? cur_air + dt : 0.f;
            if (new_detach) G(mg.last_contact_time)[k] = cur_con + dt;
            G(mg.current_contact_time)[k] = is_contact ? cur_con + dt : 0.f;
        }
        if (p < 2) stamp(2 + p);   // (diagnostic builds: the end of the first two passes)
    }
    stamp(1);
    return flag_mask;
}

}  // namespace gf
