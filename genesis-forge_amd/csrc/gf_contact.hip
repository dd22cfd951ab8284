// gf_contact.hip — Phase B2: ContactManager.step for ALL ContactManagers of a scene as one launch (the body: gf_contact_tile.h).
// gf_run_ops folds consecutive contact_step ops over the same scene arrays into one launch of this kernel — and, when the fused
// post-physics launch follows them, into THAT launch (gf_post.hip: post_step_with_contacts).
// Algorithmic traffic per env: R 8C slot ids once (+ 40 B per matched slot), W 28L (forces, mean positions, counts)
// (+ 24L link velocity / position copies when the scene provides them), RW 32L air-time state.
#include "gf_launch.h"
#include "gf_contact_tile.h"

namespace gf {

constexpr int kContactBlock = 256;
constexpr int kContactLdsBytes = 16 * 1024;

struct ContactMgr {
    ContactMgrL l;
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

__global__ __launch_bounds__(kContactBlock) void contact_kernel(const ContactMultiArgs a) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds_raw[];
    __shared__ int32_t s_with[kContactMaxMgr][GF_MAX_LINK_IDS];
    __shared__ __attribute__((aligned(8))) int32_t s_mgr[kContactMaxMgr][kContactMgrWords];
    __shared__ int32_t s_target[kContactMaxTargets];
    __shared__ uint16_t s_meta[kContactMaxTargets];   // manager | local index << 8
    const int C = a.num_contacts, T = a.total_targets, E = a.envs_per_block;
    const int64_t n0 = (int64_t)blockIdx.x * E;
    const int envs_here = (int)((int64_t)a.num_envs - n0 < E ? (int64_t)a.num_envs - n0 : E);
    const auto* kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    const ContactScene sc{a.force, a.position, a.links_quat, a.links_vel, a.links_pos, a.link_a, a.link_b, C, a.num_scene_links, T, a.dt};
    const ContactIdsPending<1> pend = contact_ids_request<1>(sc, n0, envs_here, (int)threadIdx.x, (int)blockDim.x);   // (in flight while the tables below arrive)
    // The kernel is a chain of memory round trips at the env counts of a real run (a few hundred workgroups), so the tables a
    // lane indexes with its own (env, tracked link) are staged once per workgroup — vector loads from the kernel-argument segment
    for (int i = threadIdx.x; i < kContactMaxMgr * kContactMgrWords; i += blockDim.x) {
        const int m = i / kContactMgrWords, w = i - m * kContactMgrWords;
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
    const ContactLds l = contact_lds_carve(lds_raw, E, C, reinterpret_cast<const ContactMgrL*>(&s_mgr[0][0]), s_target, s_meta, &s_with[0][0]);
    const int flag_mask = contact_tile<1, 1>(sc, l, E, n0, envs_here, (int)threadIdx.x, (int)blockDim.x, pend);   // (its first barrier covers the tables)
    // non-finite force seen: one flag per manager (contact_manager.py:399-403 prints a warning)
    for (int m = 0; m < a.num_mgr; ++m) {
        if (!a.m[m].stats) continue;
        const unsigned long long b = __ballot((flag_mask >> m) & 1);
        if (b && (threadIdx.x & (GF_WAVE - 1)) == 0) atomicOr(&stats_shard(a.m[m].stats)->contact_flags, 1);
    }
}

int validate_contact(const GfContactArgs* a) {
    if (!a || !a->contacts) return GF_E_NULL;
    if (a->num_targets <= 0 || a->num_targets > GF_MAX_LINK_IDS || a->num_with < 0 || a->num_with > GF_MAX_LINK_IDS) return GF_E_RANGE;
    if (a->num_contacts < 0 || a->num_envs < 0 || a->num_scene_links > 32767) return GF_E_RANGE;   // (slot ids are staged as 16-bit pairs)
    if (a->num_contacts > 0 && (!a->force || !a->position || !a->link_a || !a->link_b || !a->links_quat)) return GF_E_NULL;
    if (a->num_contacts > 0 && (reinterpret_cast<uintptr_t>(a->links_quat) & 15u)) return GF_E_UNSUPPORTED;
    if (a->track_air_time && (!a->last_air_time || !a->current_air_time || !a->last_contact_time || !a->current_contact_time)) return GF_E_NULL;
    return GF_OK;
}

ContactMgrL contact_mgr_image(const GfContactArgs* a) {
    ContactMgrL o{};
    o.num_targets = a->num_targets; o.num_with = a->num_with; o.has_with_filter = a->has_with_filter; o.track_air_time = a->track_air_time;
    o.air_time_threshold = a->air_time_threshold;
    o.contacts = a->contacts; o.contact_positions = a->contact_positions; o.position_counts = a->position_counts;
    o.link_vel_out = a->link_vel_out; o.link_pos_out = a->link_pos_out;
    o.last_air_time = a->last_air_time; o.current_air_time = a->current_air_time;
    o.last_contact_time = a->last_contact_time; o.current_contact_time = a->current_contact_time;
    return o;
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
        o.l = contact_mgr_image(a);
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
    // E envs per workgroup: the LDS holds their slot ids (packed pairs), masks and the relevant slots' contributions
    // (contact_lds_ints); about one (env, tracked link) pair per thread
    auto lds_bytes = [&](int e) { return (size_t)contact_lds_ints(e, C > 0 ? C : 1) * 4; };
    int E = kContactBlock / total;
    if (E > 64) E = 64;
    if (E < 1) E = 1;
    while (E > 1 && lds_bytes(E) > (size_t)kContactLdsBytes) --E;
    if (lds_bytes(E) > 48u * 1024u) return GF_E_RANGE;  // more than ~12 000 contact slots per env
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
    if (forced_e > 0 && forced_e <= 64 && lds_bytes(forced_e) <= 48u * 1024u) E = forced_e;
    k.envs_per_block = E;
    int threads = ((E * total + GF_WAVE - 1) / GF_WAVE) * GF_WAVE;
    if (threads > kContactBlock) threads = kContactBlock;
    const unsigned grid = (unsigned)(((int64_t)a0->num_envs + E - 1) / E);
    const size_t lds = lds_bytes(E);
    PhaseScope scope(GF_PHASE_CONTACT, s);
    GF_LAUNCH(scope, contact_kernel, grid, threads, lds, s, k);
    return launch_status();
}

}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_contact_step(const GfContactArgs* a, void* stream) {
    return gf::contact_launch(&a, 1, (hipStream_t)stream);
}
