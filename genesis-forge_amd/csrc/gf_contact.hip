// gf_contact.hip — Phase B2: ContactManager.step as one launch.
//
// Replaces, per ContactManager (the gait config has three):
//   managers/contact/contact_manager.py:399-403   isnan/isinf .any() (2 host syncs) + nan_to_num
//   managers/contact/contact_manager.py:408-411   3 fill_ launches
//   managers/contact/kernel.py:35-90              the Taichi kernel (atomic += over (env, contact, target))
//   managers/contact/contact_manager.py:434-477   norm > threshold + 4 torch.where air-time updates
//
// One lane per (env, target link) pair: the lane walks that env's C contact slots in slot order and
// keeps its force / position / count accumulators in registers, so there are no atomics at all and
// the f32 sum order is fixed (the Taichi reference's atomic order is not).  The L lanes of one env
// read the same link_a/link_b words (broadcast) and consecutive envs are adjacent in memory.  The
// force-norm / threshold / air-time state update that the reference runs afterwards as ~12 separate
// launches is done by the same lane while the summed force is still in registers.
// Algorithmic traffic per env: R C*(8+12+12) contact slots (+16 per matched slot for the link
// quaternion), W 28L (forces, mean positions, counts), RW 32L air-time state.
#include "gf_launch.h"

namespace gf {

__global__ __launch_bounds__(kEnvBlock) void contact_kernel(const GfContactArgs a) {
    const int64_t gid = (int64_t)blockIdx.x * kEnvBlock + threadIdx.x;
    const int L = a.num_targets, C = a.num_contacts, W = a.num_with;
    const int64_t n = gid / L;
    const int t = (int)(gid - n * L);
    const bool live = n < a.num_envs;
    int flag = 0;
    if (live) {
        const int target = a.target_link_ids[t];
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f, cnt = 0.f;
        const int32_t* la_row = a.link_a + n * C;
        const int32_t* lb_row = a.link_b + n * C;
        for (int c = 0; c < C; ++c) {
            const int la = la_row[c], lb = lb_row[c];
            const bool is_a = la == target, is_b = lb == target;
            if (!(is_a || is_b)) continue;
            bool include = true;
            if (a.has_with_filter) {
                include = false;
                for (int w = 0; w < W; ++w) {
                    const int wl = a.with_link_ids[w];
                    if ((is_a && lb == wl) || (is_b && la == wl)) { include = true; break; }
                }
            }
            if (!include) continue;
            const float* fr = a.force + (n * C + c) * 3;
            const float* pr = a.position + (n * C + c) * 3;
            float fx = fr[0], fy = fr[1], fz = fr[2];
            // torch.nan_to_num(force, nan=0, posinf=0, neginf=0)   contact_manager.py:401-403
            if (isnan(fx) || isinf(fx)) { fx = 0.f; flag = 1; }
            if (isnan(fy) || isinf(fy)) { fy = 0.f; flag = 1; }
            if (isnan(fz) || isinf(fz)) { fz = 0.f; flag = 1; }
            p0 += pr[0]; p1 += pr[1]; p2 += pr[2];
            cnt += 1.0f;
            // force is expressed on link_b; on link_a it is the reaction (kernel.py:74-78)
            const int ql = is_b ? lb : la;
            const float4 q = load_quat(a.links_quat, n * a.num_scene_links + ql);
            const V3 r = is_b ? rot_inv(q, V3{fx, fy, fz}) : rot_inv(q, V3{-fx, -fy, -fz});
            f0 += r.x; f1 += r.y; f2 += r.z;
        }
        const int64_t k = n * L + t;
        a.contacts[3 * k + 0] = f0;
        a.contacts[3 * k + 1] = f1;
        a.contacts[3 * k + 2] = f2;
        if (a.contact_positions) {  // kernel.py:84-90
            a.contact_positions[3 * k + 0] = cnt > 0.f ? p0 / cnt : p0;
            a.contact_positions[3 * k + 1] = cnt > 0.f ? p1 / cnt : p1;
            a.contact_positions[3 * k + 2] = cnt > 0.f ? p2 / cnt : p2;
        }
        if (a.position_counts) a.position_counts[k] = cnt;
        if (a.links_vel && a.link_vel_out) {  // compact per-manager copy of the tracked links' velocities (feet_slide)
            const float* sv = a.links_vel + (n * a.num_scene_links + target) * 3;
            a.link_vel_out[3 * k + 0] = sv[0];
            a.link_vel_out[3 * k + 1] = sv[1];
            a.link_vel_out[3 * k + 2] = sv[2];
        }
        if (a.links_pos && a.link_pos_out) {  // … and of their positions (the gait manager's foot_height_reward)
            const float* sp = a.links_pos + (n * a.num_scene_links + target) * 3;
            a.link_pos_out[3 * k + 0] = sp[0];
            a.link_pos_out[3 * k + 1] = sp[1];
            a.link_pos_out[3 * k + 2] = sp[2];
        }
        if (a.track_air_time) {  // contact_manager.py:441-477
            const float dt = a.dt;
            const bool is_contact = norm3(f0, f1, f2) > a.air_time_threshold;
            const float cur_air = a.current_air_time[k], cur_con = a.current_contact_time[k];
            const bool new_contact = (cur_air > 0.f) && is_contact;
            const bool new_detach = (cur_con > 0.f) && !is_contact;
            if (new_contact) a.last_air_time[k] = cur_air + dt;
            a.current_air_time[k] = !is_contact ? cur_air + dt : 0.f;
            if (new_detach) a.last_contact_time[k] = cur_con + dt;
            a.current_contact_time[k] = is_contact ? cur_con + dt : 0.f;
        }
    }
    if (a.stats) {
        const unsigned long long m = __ballot(flag);
        if (m && threadIdx.x == 0) atomicOr(&stats_shard(a.stats)->contact_flags, 1);
    }
}

}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_contact_step(const GfContactArgs* a, void* stream) {
    if (!a || !a->contacts) return GF_E_NULL;
    if (a->num_targets <= 0 || a->num_targets > GF_MAX_LINK_IDS || a->num_with < 0 || a->num_with > GF_MAX_LINK_IDS) return GF_E_RANGE;
    if (a->num_contacts < 0 || a->num_envs < 0) return GF_E_RANGE;
    if (a->num_contacts > 0 && (!a->force || !a->position || !a->link_a || !a->link_b || !a->links_quat)) return GF_E_NULL;
    if (a->num_contacts > 0 && (reinterpret_cast<uintptr_t>(a->links_quat) & 15u)) return GF_E_UNSUPPORTED;
    if (a->track_air_time && (!a->last_air_time || !a->current_air_time || !a->last_contact_time || !a->current_contact_time)) return GF_E_NULL;
    if (a->num_envs == 0) return GF_OK;
    hipStream_t s = (hipStream_t)stream;
    gf::PhaseScope scope(GF_PHASE_CONTACT, s);
    scope.begin_bracket();
    gf::contact_kernel<<<gf::env_grid((int64_t)a->num_envs * a->num_targets), gf::kEnvBlock, 0, s>>>(*a);
    return gf::launch_status();
}
