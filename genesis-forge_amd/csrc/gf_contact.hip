// gf_contact.hip — Phase B2: ContactManager.step for ALL ContactManagers of a scene as one launch.
//
// Replaces, per ContactManager (the gait config has three):
//   managers/contact/contact_manager.py:399-403   isnan/isinf .any() (2 host syncs) + nan_to_num
//   managers/contact/contact_manager.py:408-411   3 fill_ launches
//   managers/contact/kernel.py:35-90              the Taichi kernel (atomic += over (env, contact, target))
//   managers/contact/contact_manager.py:434-477   norm > threshold + 4 torch.where air-time updates
//
// Layout.  A workgroup owns E consecutive envs.  (1) Their link_a / link_b rows (E·C ints each, contiguous in memory) are
// staged into LDS with flat coalesced loads — every contact slot id is read from HBM exactly once, whatever the number of
// managers and tracked links — and every slot that holds a contact (ids >= 0; contacts are sparse) sets its bit in the
// env's occupancy mask (an LDS OR: order independent).  (2) One lane per (env, tracked link of any manager) walks ONLY the
// occupied slots of its env, in slot order (ctz over the mask), with its force / position / count accumulators in
// registers: no float atomics, a fixed f32 sum order (the Taichi reference's atomic order is not).  Force and position are
// fetched for matching slots only; the only quaternion a lane can need is its own target link's (a matching slot always
// involves the target), loaded up front together with the air-time state.
// The force-norm / threshold / air-time update that the reference runs afterwards as ~12 separate launches is done by
// the same lane while the summed force is in registers.
// gf_run_ops folds consecutive contact_step ops over the same scene arrays into one launch of this kernel.
// Algorithmic traffic per env: R 8C slot ids once (+ 40 B per matched slot), W 28L (forces, mean positions, counts)
// (+ 24L link velocity / position copies when the scene provides them), RW 32L air-time state.
#include "gf_launch.h"

namespace gf {

constexpr int kContactMaxMgr = 4;
constexpr int kContactMaxTargets = 64;   // tracked links over all managers of one launch
constexpr int kContactBlock = 256;
constexpr int kContactLdsBytes = 16 * 1024;
#define GF_CONTACT_INLINE __attribute__((always_inline))
typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));   // dword aligned; one global_{load,store}_dwordx3: a wave moves 768 contiguous bytes
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st3(float* p, float x, float y, float z) { *reinterpret_cast<GF_GLOBAL f32x3*>(G(p)) = f32x3{x, y, z}; }
__device__ __forceinline__ V3 ld3(const float* p) {
    const f32x3 v = *reinterpret_cast<const GF_GLOBAL f32x3*>(G(p));
    return V3{v.x, v.y, v.z};
}

struct ContactMgr {
    int32_t num_targets, num_with, has_with_filter, track_air_time;
    float air_time_threshold;
    int32_t _pad;
    float *contacts, *contact_positions, *position_counts, *link_vel_out, *link_pos_out;
    float *last_air_time, *current_air_time, *last_contact_time, *current_contact_time;
    GfStepStats* stats;
    int32_t with_link_ids[GF_MAX_LINK_IDS];
};

struct ContactMultiArgs {
    int32_t num_envs, num_contacts, num_scene_links, num_mgr;
    int32_t total_targets, envs_per_block;
    float dt;
    int32_t _pad;
    const float *force, *position, *links_quat, *links_vel, *links_pos;
    const int32_t *link_a, *link_b;
    int32_t target_ids[kContactMaxTargets];
    uint8_t mgr_of[kContactMaxTargets];
    uint8_t local_of[kContactMaxTargets];
    ContactMgr m[kContactMaxMgr];
};
static_assert(sizeof(ContactMultiArgs) <= 4096, "kernarg segment");

// floor(i / d) by multiply-shift, exact for i < 2^40 / d (i < 64·C here)
struct FastDivC {
    uint64_t m;
    uint32_t d;
    __device__ __forceinline__ explicit FastDivC(int div) : m(div > 1 ? ((1ull << 40) + (uint64_t)div - 1ull) / (uint64_t)div : 0ull), d((uint32_t)div) {}
    __device__ __forceinline__ int div(int i) const { return d > 1 ? (int)(((uint64_t)(uint32_t)i * m) >> 40) : i; }
};

__global__ __launch_bounds__(kContactBlock) void contact_kernel(const ContactMultiArgs a) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds_raw[];
    __shared__ int32_t s_with[kContactMaxMgr][GF_MAX_LINK_IDS];
    // The per-manager fields and the target table, staged once per workgroup: a lane's manager / tracked link is lane-dependent, and
    // picking 14 fields x 4 managers (and 3 x 16 table entries) with constant-index select chains out of SGPRs cost ~250 of the
    // kernel's ~1 100 instructions per wave and 127 spilled SGPRs — on a kernel PMC counters show to be issue-bound (r02_i).
    constexpr int kMgrWords = (int)(offsetof(ContactMgr, with_link_ids) / 4);   // the scalar fields and pointers of one manager
    __shared__ __attribute__((aligned(8))) int32_t s_mgr[kContactMaxMgr][kMgrWords];
    __shared__ int32_t s_target[kContactMaxTargets];
    __shared__ uint16_t s_meta[kContactMaxTargets];   // manager | local index << 8
    const int C = a.num_contacts, T = a.total_targets, E = a.envs_per_block;
    const int MW = (C + 31) >> 5;  // 32-bit words of one env's "slot holds a contact" mask
    const int64_t n0 = (int64_t)blockIdx.x * E;
    const int envs_here = (int)((int64_t)a.num_envs - n0 < E ? (int64_t)a.num_envs - n0 : E);
    const int slots = envs_here * C;
    const int pairs = envs_here * T;
    int32_t* sa = lds_raw;                                                  // [E*C] link_a
    int32_t* sb = lds_raw + E * C;                                          // [E*C] link_b
    uint32_t* smask = reinterpret_cast<uint32_t*>(lds_raw + 2 * E * C);     // [E][MW]
    const auto* kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();

    // The kernel is a chain of memory round trips at the env counts of a real run (a few hundred workgroups), so everything
    // that does not depend on an earlier load goes out first: this lane's (env, tracked link) row of the target table — a
    // vector load from the kernel-argument segment — together with the slot ids and the with-filter tables.
    const FastDivC dT(T);
    auto pair_meta = [&](int pr, int& e, int& t, int& target, int& mi, int& lt) GF_CONTACT_INLINE {
        const bool live = pr < pairs;
        e = live ? dT.div(pr) : 0;
        t = live ? pr - e * T : 0;
        target = s_target[t];
        const int meta = s_meta[t];
        mi = meta & 0xff;
        lt = meta >> 8;
    };
    for (int i = threadIdx.x; i < kContactMaxMgr * kMgrWords; i += blockDim.x) {
        const int m = i / kMgrWords, w = i - m * kMgrWords;
        (&s_mgr[0][0])[i] = *(const __attribute__((address_space(4))) int32_t*)(kp + offsetof(ContactMultiArgs, m) + (size_t)m * sizeof(ContactMgr) + 4 * w);
    }
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        s_target[i] = *(const __attribute__((address_space(4))) int32_t*)(kp + offsetof(ContactMultiArgs, target_ids) + 4 * i);
        s_meta[i] = (uint16_t)(*(const __attribute__((address_space(4))) uint8_t*)(kp + offsetof(ContactMultiArgs, mgr_of) + i) |
                               (*(const __attribute__((address_space(4))) uint8_t*)(kp + offsetof(ContactMultiArgs, local_of) + i) << 8));
    }
    for (int i = threadIdx.x; i < kContactMaxMgr * GF_MAX_LINK_IDS; i += blockDim.x) {
        const int m = i / GF_MAX_LINK_IDS, w = i - m * GF_MAX_LINK_IDS;
        (&s_with[0][0])[i] = *(const __attribute__((address_space(4))) int32_t*)(kp + offsetof(ContactMultiArgs, m) + (size_t)m * sizeof(ContactMgr) +
                                                                                 offsetof(ContactMgr, with_link_ids) + 4 * w);
    }
    for (int i = threadIdx.x; i < envs_here * MW; i += blockDim.x) smask[i] = 0u;
    __syncthreads();
    {   // phase 1: the E rows of slot ids are contiguous — flat coalesced copy; slots that hold a contact set their bit
        // (an OR: the mask does not depend on the order the lanes arrive in)
        const GF_GLOBAL int32_t* ga = G(a.link_a) + n0 * C;
        const GF_GLOBAL int32_t* gb = G(a.link_b) + n0 * C;
        const FastDivC dc(C);
        auto mark = [&](int i, int la, int lb) GF_CONTACT_INLINE {
            if (la >= 0 || lb >= 0) {
                const int e = dc.div(i), c = i - e * C;
                atomicOr(&smask[e * MW + (c >> 5)], 1u << (c & 31));
            }
        };
        // four ids per lane and array (dwordx4: 1 KiB per wave instruction) when the block's rows start 16-byte aligned
        const bool vec = ((E * C) & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.link_a) | reinterpret_cast<uintptr_t>(a.link_b)) & 15u) == 0;
        const int slots4 = vec ? (slots >> 2) : 0;
        for (int i4 = threadIdx.x; i4 < slots4; i4 += blockDim.x) {
            const i32x4 va = reinterpret_cast<const GF_GLOBAL i32x4*>(ga)[i4], vb = reinterpret_cast<const GF_GLOBAL i32x4*>(gb)[i4];
            reinterpret_cast<i32x4*>(sa)[i4] = va;
            reinterpret_cast<i32x4*>(sb)[i4] = vb;
            // contacts are sparse: one (rarely taken) branch per four slots instead of four
            const int occ4 = ((va.x & vb.x) >= 0 ? 1 : 0) | ((va.y & vb.y) >= 0 ? 2 : 0) | ((va.z & vb.z) >= 0 ? 4 : 0) | ((va.w & vb.w) >= 0 ? 8 : 0);
            if (occ4) {
                const int i = i4 << 2;
                if (occ4 & 1) mark(i, va.x, vb.x);
                if (occ4 & 2) mark(i + 1, va.y, vb.y);
                if (occ4 & 4) mark(i + 2, va.z, vb.z);
                if (occ4 & 8) mark(i + 3, va.w, vb.w);
            }
        }
        for (int i = (slots4 << 2) + threadIdx.x; i < slots; i += blockDim.x) {
            const int la = ga[i], lb = gb[i];
            sa[i] = la; sb[i] = lb;
            mark(i, la, lb);
        }
    }
    // phase 2: one lane per (env, tracked link) walks ONLY the occupied slots of its env, in slot order
    int flag_mask = 0;  // bit m: this lane sanitised a non-finite force for manager m
    for (int pr0 = 0; pr0 < pairs; pr0 += blockDim.x) {
        const int pr = pr0 + (int)threadIdx.x;
        const bool live = pr < pairs;
        int e, t, target, mi, lt;
        pair_meta(pr, e, t, target, mi, lt);
        const ContactMgr& mg = *reinterpret_cast<const ContactMgr*>(&s_mgr[mi][0]);   // only the fields in front of with_link_ids
        const int64_t n = n0 + e;
        const int L = mg.num_targets, W = mg.num_with;
        const int64_t k = n * L + lt;
        // loop-invariant loads first: a matching slot always involves the target link itself, so its quaternion is the only
        // one this lane can need (kernel.py:74-78); the air-time state is read before the scan as well
        float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
        if (live && C > 0) q = ldg4(G(a.links_quat) + (n * a.num_scene_links + target) * 4);
        float cur_air = 0.f, cur_con = 0.f;
        if (live && mg.track_air_time) { cur_air = G(mg.current_air_time)[k]; cur_con = G(mg.current_contact_time)[k]; }
        V3 lvel{0.f, 0.f, 0.f}, lpos{0.f, 0.f, 0.f};
        const bool copy_vel = live && a.links_vel && mg.link_vel_out, copy_pos = live && a.links_pos && mg.link_pos_out;
        if (copy_vel) lvel = ld3(a.links_vel + 3 * (n * a.num_scene_links + target));
        if (copy_pos) lpos = ld3(a.links_pos + 3 * (n * a.num_scene_links + target));
        if (pr0 == 0) __syncthreads();  // phase 1's LDS writes
        if (!live) continue;
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, cnt = 0.f;
        const int32_t* la_row = sa + e * C;
        const int32_t* lb_row = sb + e * C;
        for (int wd = 0; wd < MW; ++wd) {
            uint32_t bits = smask[e * MW + wd];
            while (bits) {
                const int c = (wd << 5) + __builtin_ctz(bits);
                bits &= bits - 1u;
                const int la = la_row[c], lb = lb_row[c];
                const bool is_a = la == target, is_b = lb == target;
                if (!(is_a || is_b)) continue;
                bool include = true;
                if (mg.has_with_filter) {
                    include = false;
                    for (int w = 0; w < W; ++w) {
                        const int wl = s_with[mi][w];
                        if ((is_a && lb == wl) || (is_b && la == wl)) { include = true; break; }
                    }
                }
                if (!include) continue;
                const V3 fv = ld3(a.force + (n * C + c) * 3), pv = ld3(a.position + (n * C + c) * 3);
                float fx = fv.x, fy = fv.y, fz = fv.z;
                const float px = pv.x, py = pv.y, pz = pv.z;
                // torch.nan_to_num(force, nan=0, posinf=0, neginf=0)   contact_manager.py:401-403
                if (isnan(fx) || isinf(fx)) { fx = 0.f; flag_mask |= 1 << mi; }
                if (isnan(fy) || isinf(fy)) { fy = 0.f; flag_mask |= 1 << mi; }
                if (isnan(fz) || isinf(fz)) { fz = 0.f; flag_mask |= 1 << mi; }
                p0 += px; p1 += py; p2 += pz;
                cnt += 1.0f;
                // force is expressed on link_b; on link_a it is the reaction (kernel.py:74-78)
                const V3 r = is_b ? rot_inv(q, V3{fx, fy, fz}) : rot_inv(q, V3{-fx, -fy, -fz});
                f0 += r.x; f1 += r.y; f2 += r.z;
            }
        }
        // every load of this pair has been consumed or was issued before the scan: say so once, or the compiler — loads and stores
        // share vmcnt and complete out of order with respect to each other — puts waits between the conditional stores below
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        st3(mg.contacts + 3 * k, f0, f1, f2);
        if (mg.contact_positions)  // kernel.py:84-90
            st3(mg.contact_positions + 3 * k, cnt > 0.f ? p0 / cnt : p0, cnt > 0.f ? p1 / cnt : p1, cnt > 0.f ? p2 / cnt : p2);
        if (mg.position_counts) G(mg.position_counts)[k] = cnt;
        if (copy_vel) st3(mg.link_vel_out + 3 * k, lvel.x, lvel.y, lvel.z);  // compact per-manager copy of the tracked links' velocities (feet_slide)
        if (copy_pos) st3(mg.link_pos_out + 3 * k, lpos.x, lpos.y, lpos.z);  // … and of their positions (the gait manager's foot_height_reward)
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
    }
    // non-finite force seen: one flag per manager (contact_manager.py:399-403 prints a warning)
    for (int m = 0; m < a.num_mgr; ++m) {
        if (!a.m[m].stats) continue;
        const unsigned long long b = __ballot((flag_mask >> m) & 1);
        if (b && (threadIdx.x & (GF_WAVE - 1)) == 0) atomicOr(&stats_shard(a.m[m].stats)->contact_flags, 1);
    }
}

static int validate_contact(const GfContactArgs* a) {
    if (!a || !a->contacts) return GF_E_NULL;
    if (a->num_targets <= 0 || a->num_targets > GF_MAX_LINK_IDS || a->num_with < 0 || a->num_with > GF_MAX_LINK_IDS) return GF_E_RANGE;
    if (a->num_contacts < 0 || a->num_envs < 0) return GF_E_RANGE;
    if (a->num_contacts > 0 && (!a->force || !a->position || !a->link_a || !a->link_b || !a->links_quat)) return GF_E_NULL;
    if (a->num_contacts > 0 && (reinterpret_cast<uintptr_t>(a->links_quat) & 15u)) return GF_E_UNSUPPORTED;
    if (a->track_air_time && (!a->last_air_time || !a->current_air_time || !a->last_contact_time || !a->current_contact_time)) return GF_E_NULL;
    return GF_OK;
}

// Managers that can share one launch read the same scene arrays (the usual case: every ContactManager of an env).
bool contact_compatible(const GfContactArgs* x, const GfContactArgs* y) {
    return x->num_envs == y->num_envs && x->num_contacts == y->num_contacts && x->num_scene_links == y->num_scene_links &&
           x->force == y->force && x->position == y->position && x->link_a == y->link_a && x->link_b == y->link_b &&
           x->links_quat == y->links_quat && x->links_vel == y->links_vel && x->links_pos == y->links_pos && x->dt == y->dt;
}

int contact_launch(const GfContactArgs* const* mgrs, int num, hipStream_t s) {
    if (num < 1 || num > kContactMaxMgr) return GF_E_RANGE;
    ContactMultiArgs k{};
    int total = 0;
    for (int m = 0; m < num; ++m) {
        const GfContactArgs* a = mgrs[m];
        const int rc = validate_contact(a);
        if (rc) return rc;
        if (m > 0 && !contact_compatible(mgrs[0], a)) return GF_E_UNSUPPORTED;
        if (total + a->num_targets > kContactMaxTargets) return GF_E_RANGE;
        ContactMgr& o = k.m[m];
        o.num_targets = a->num_targets; o.num_with = a->num_with; o.has_with_filter = a->has_with_filter; o.track_air_time = a->track_air_time;
        o.air_time_threshold = a->air_time_threshold;
        o.contacts = a->contacts; o.contact_positions = a->contact_positions; o.position_counts = a->position_counts;
        o.link_vel_out = a->link_vel_out; o.link_pos_out = a->link_pos_out;
        o.last_air_time = a->last_air_time; o.current_air_time = a->current_air_time;
        o.last_contact_time = a->last_contact_time; o.current_contact_time = a->current_contact_time;
        o.stats = a->stats;
        for (int w = 0; w < a->num_with; ++w) o.with_link_ids[w] = a->with_link_ids[w];
        for (int t = 0; t < a->num_targets; ++t) {
            k.target_ids[total] = a->target_link_ids[t];
            k.mgr_of[total] = (uint8_t)m;
            k.local_of[total] = (uint8_t)t;
            ++total;
        }
    }
    const GfContactArgs* a0 = mgrs[0];
    if (a0->num_envs == 0) return GF_OK;
    k.num_envs = a0->num_envs; k.num_contacts = a0->num_contacts; k.num_scene_links = a0->num_scene_links; k.num_mgr = num;
    k.total_targets = total; k.dt = a0->dt;
    k.force = a0->force; k.position = a0->position; k.links_quat = a0->links_quat; k.links_vel = a0->links_vel; k.links_pos = a0->links_pos;
    k.link_a = a0->link_a; k.link_b = a0->link_b;
    const int C = a0->num_contacts;
    // E envs per workgroup: 8 B of LDS per contact slot (the two ids) + the occupancy mask; ≈ one (env, link) pair per thread
    const int per_env = (C > 0 ? C : 1) * 8 + ((C + 31) / 32) * 4;
    int E = kContactBlock / total;
    if (E > 64) E = 64;
    if (E < 1) E = 1;
    if (E > kContactLdsBytes / per_env) E = kContactLdsBytes / per_env;
    if (E < 1) return GF_E_RANGE;  // more than ~2 000 contact slots per env
    // small problems: keep at least ~2 workgroups per CU busy rather than 64-env tiles on a quarter of the chip
    const int e_max = E;
    while (E > 8 && ((int64_t)a0->num_envs + E - 1) / E < 512) E >>= 1;
    // A block's rows of slot ids should start and end on cache-line boundaries (E·C·4 bytes a multiple of 128): a line shared by two
    // blocks is fetched by both.  Take the largest such E that fits a block and still leaves 512 workgroups, unless it is much smaller
    // than the choice above.  Measured over E on one box (profiles/r03_w_contact_e.txt, r03_w_contact_e_humanoid.txt): gait task
    // (C = 60: multiples of 8) 19 → 16 at 65 536 envs and 9 → 16 at 8 192, whole step 136.4 → 133.5 and 36.7 → 35.5 µs; humanoid
    // configs (C = 30: multiples of 16) 14 → 16, 24.5 → 24.1 and 48.0 → 44.4 µs — E = 16 is the minimum of every sweep.
    if (C > 0) {
        int q = 32, c = C;
        while (q > 1 && (c & 1) == 0) { q >>= 1; c >>= 1; }   // q = 32 / gcd(C, 32)
        int al = (e_max / q) * q;
        while (al > q && ((int64_t)a0->num_envs + al - 1) / al < 512) al -= q;
        if (al >= 1 && 4 * al >= 3 * E && ((int64_t)a0->num_envs + al - 1) / al >= 512) E = al;
    }
    static const int forced_e = getenv("GF_CONTACT_E") ? atoi(getenv("GF_CONTACT_E")) : 0;   // experiments only
    if (forced_e > 0 && forced_e <= kContactLdsBytes / per_env) E = forced_e;
    k.envs_per_block = E;
    int threads = ((E * total + GF_WAVE - 1) / GF_WAVE) * GF_WAVE;
    if (threads > kContactBlock) threads = kContactBlock;
    const unsigned grid = (unsigned)(((int64_t)a0->num_envs + E - 1) / E);
    const size_t lds = (size_t)E * per_env;
    PhaseScope scope(GF_PHASE_CONTACT, s);
    GF_LAUNCH(scope, contact_kernel, grid, threads, lds, s, k);
    return launch_status();
}

}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_contact_step(const GfContactArgs* a, void* stream) {
    return gf::contact_launch(&a, 1, (hipStream_t)stream);
}
