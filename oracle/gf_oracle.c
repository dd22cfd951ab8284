/*
 * gf_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, scalar, one-env-at-a-time restatement of the per-tick manager work of the
 * reference's ManagedEnvironment.step() (/root/reference/genesis_forge/managed_env.py:274-334).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path (genesis-forge_amd/csrc/*.hip behind include/gf_step.h) never does.
 *
 * Parity status: PINNED against the reference itself — tests/golden/*.npz were generated in the
 * build container by importing /root/reference/genesis_forge (with stub genesis/gstaichi/
 * gymnasium/tensordict modules, tools/gen_golden.py) and recording its inputs/outputs; this
 * oracle is checked against every one of them on the CPU (`pytest -m "not gpu"`): tests/test_terms_golden.py (every mdp term,
 * action managers, air time, orientation sweep, entity / observation getters), tests/test_golden_trajectory.py and
 * tests/test_examples.py (whole trajectories, incl. the reference's six example files), tests/test_contact_kernel.py,
 * tests/test_terrain.py.
 * UNPINNED sub-parts (third-party code absent from /root/reference):
 *   - genesis.utils.geom.transform_by_quat / inv_quat (genesis-world>=0.3.4): restated from
 *     the mathematical definition  v' = v + w*t + qv x t,  t = 2*(qv x v),  inv = conjugate;
 *   - gstaichi's JIT semantics for kernel_get_contact_forces (the order of its atomic `+=`): the fixture was recorded by
 *     executing the reference's kernel SOURCE under a serial emulation of ti.ndrange / ti.Vector / ti.static
 *     (tools/ref_stubs.py), accumulation in contact-slot order (managers/contact/kernel.py:35-90);
 *   - torch's RNG stream: draws are inputs (dense U[0,1) arrays) or Philox4x32-10.
 *
 * It shares the POD descriptors of include/gf_step.h (host pointers instead of device
 * pointers); every gfo_* function mirrors the gf_* entry point of the same name.
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  All arithmetic is
 * f32 with one rounding per operation (no FMA contraction), in the operation order of the
 * reference's torch expressions.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/gf_step.h"

#define GFO_EXPORT __attribute__((visibility("default")))

/* ---------------------------------------------------------------- Philox4x32-10 ---------- */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                 uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* U[0,1) with 24 random bits for (env, column) of (seed, stream). */
static inline float philox_uniform(uint64_t seed, uint64_t stream, uint32_t env, uint32_t col) {
    uint32_t x[4];
    philox4x32_10(env, col >> 2, (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), x);
    return (float)(x[col & 3] >> 8) * 5.9604644775390625e-8f; /* 2^-24 */
}

/* row stride of a command view in floats (0 = dense) */
static inline int64_t cmd_stride(const GfCommandView* c) { return c->stride ? c->stride : c->width; }

/* torch's `%` on float tensors (aten remainder kernel): fmod, then moved to the divisor's sign */
static inline float torch_remainder(float a, float b) {
    float m = fmodf(a, b);
    if ((m != 0.0f) && ((b < 0.0f) != (m < 0.0f))) m += b;
    return m;
}

/* swing (bit 0) / stance (bit 1) of one foot, examples/gait_trainer/gait_command_manager.py:331-338 */
static inline int gait_foot_flags(float phase, float offset, float two_pi, float pi) {
    const float phi = torch_remainder(phase + offset, 1.0f) * two_pi;
    return (((phi >= 0.0f) && (phi < pi)) ? 1 : 0) | (((phi >= pi) && (phi < two_pi)) ? 2 : 0);
}

static inline float draw_u(const float* draws, int64_t idx, uint64_t seed, uint64_t stream, uint32_t env, uint32_t col) {
    return draws ? draws[idx] : philox_uniform(seed, stream, env, col);
}

/* uniform_(lo, hi) as torch's uniform_real does it: x*(to-from)+from  (f32, two roundings) */
static inline float uniform_range(float u, float lo, float hi) { return u * (hi - lo) + lo; }

/* ---------------------------------------------------------------- quaternion helpers ----- */
/* transform_by_quat(v, inv_quat(q)): rotate a world vector into the body frame
 * (genesis_forge/utils.py:13-55; managers/entity_manager.py:130-146,195). */
static inline void rot_inv(const float* q, const float* v, float* o) {
    const float w = q[0], a = -q[1], b = -q[2], c = -q[3]; /* inv_quat = conjugate */
    const float t0 = (b * v[2] - c * v[1]) * 2.0f;
    const float t1 = (c * v[0] - a * v[2]) * 2.0f;
    const float t2 = (a * v[1] - b * v[0]) * 2.0f;
    o[0] = (v[0] + w * t0) + (b * t2 - c * t1);
    o[1] = (v[1] + w * t1) + (c * t0 - a * t2);
    o[2] = (v[2] + w * t2) + (a * t1 - b * t0);
}

static inline float norm3(const float* v) { return sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
static inline float norm2(float x, float y) { return sqrtf(x * x + y * y); }

static inline void body_lin_vel(const GfEntityView* e, int64_t n, float* o) { rot_inv(e->quat + 4 * n, e->lin_vel + 3 * n, o); }
static inline void body_ang_vel(const GfEntityView* e, int64_t n, float* o) { rot_inv(e->quat + 4 * n, e->ang_vel + 3 * n, o); }
static inline void proj_gravity(const GfEntityView* e, int64_t n, float* o) {
    const float g[3] = {0.0f, 0.0f, -1.0f};
    rot_inv(e->quat + 4 * n, g, o);
}

/* ---------------------------------------------------------------- terrain ---------------- */
/* TerrainManager.get_terrain_height for one point (managers/terrain_manager.py:100-166).
 *   :117-136  norm = 2*(v - v_min)/(v_max - v_min) - 1 as the in-place chain sub_, div_, mul_(2), sub_(1)
 *   :153-159  F.grid_sample(field[1,H,W], grid(x->W, y->H), bilinear, border, align_corners=True):
 *             unnormalise ((c+1)/2)*(size-1), clip to [0,size-1], floor, the four corner weights, taps outside the field
 *             contribute 0 (ATen GridSampler: grid_sampler_compute_source_index / bilinear taps nw,ne,sw,se)
 *   :112-114  no height field: the terrain origin's z. */
static float terrain_height(const GfTerrainView* tv, float x, float y) {
    if (!tv->height_field) return tv->origin_z;
    float nx = x - tv->x_min;
    nx = nx / tv->x_span;
    nx = nx * 2.0f;
    nx = nx - 1.0f;
    float ny = y - tv->y_min;
    ny = ny / tv->y_span;
    ny = ny * 2.0f;
    ny = ny - 1.0f;
    const int W = tv->cols, H = tv->rows;
    float ix = ((nx + 1.0f) / 2.0f) * (float)(W - 1);
    float iy = ((ny + 1.0f) / 2.0f) * (float)(H - 1);
    if (ix < 0.0f) ix = 0.0f;
    if (ix > (float)(W - 1)) ix = (float)(W - 1);
    if (iy < 0.0f) iy = 0.0f;
    if (iy > (float)(H - 1)) iy = (float)(H - 1);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
    const float nw = (fx1 - ix) * (fy1 - iy);
    const float ne = (ix - fx0) * (fy1 - iy);
    const float sw = (fx1 - ix) * (iy - fy0);
    const float se = (ix - fx0) * (iy - fy0);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float* f = tv->height_field;
    const float v_nw = f[y0 * W + x0];
    const float v_ne = x1 < W ? f[y0 * W + x1] : 0.0f;
    const float v_sw = y1 < H ? f[y1 * W + x0] : 0.0f;
    const float v_se = (x1 < W && y1 < H) ? f[y1 * W + x1] : 0.0f;
    return ((v_nw * nw + v_ne * ne) + v_sw * sw) + v_se * se;
}

GFO_EXPORT int gfo_terrain_height(const GfTerrainHeightArgs* a) {
    if (!a || !a->out) return GF_E_NULL;
    if (a->num < 0) return GF_E_RANGE;
    if (a->terrain.height_field) {
        if (!a->x || !a->y) return GF_E_NULL;
        if (a->terrain.rows < 1 || a->terrain.cols < 1 || a->x_stride < 0 || a->y_stride < 0) return GF_E_RANGE;
    }
    for (int64_t i = 0; i < a->num; ++i)
        a->out[i] = a->terrain.height_field ? terrain_height(&a->terrain, a->x[i * a->x_stride], a->y[i * a->y_stride]) : a->terrain.origin_z;
    return GF_OK;
}

/* sin / cos as a fixed f32 operation sequence (the library's arithmetic contract for the spawn quaternion: the same
 * sequence runs on the GPU, so both sides produce the same bits; within 2 ulp of libm, i.e. of the reference's torch.sin/cos):
 * k = rint(x*2/pi), three-part Cody-Waite reduction r = x - k*pi/2, Cephes minimax polynomials on [-pi/4, pi/4]. */
static void sincos_det(float x, float* s, float* c) {
    const float k = rintf(x * 0.63661977236758134f);
    float r = x - k * 1.5703125f;
    r = r - k * 4.837512969970703125e-4f;
    r = r - k * 7.54978995489188216e-8f;
    const float z = r * r;
    const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    const int q = (int)k & 3;
    const float sv = (q & 1) ? pc : ps, cv = (q & 1) ? ps : pc;
    *s = (q & 2) ? -sv : sv;
    *c = ((q + 1) & 2) ? -cv : cv;
}

/* genesis.utils.geom.xyz_to_quat (third-party, absent; call sites mdp/reset.py:63,194): extrinsic x-y-z Euler -> (w,x,y,z) */
static void xyz_to_quat(float ax, float ay, float az, float* q) {
    float sx, cx, sy, cy, sz, cz;
    sincos_det(ax * 0.5f, &sx, &cx);
    sincos_det(ay * 0.5f, &sy, &cy);
    sincos_det(az * 0.5f, &sz, &cz);
    q[0] = (cx * cy) * cz + (sx * sy) * sz;
    q[1] = (sx * cy) * cz - (cx * sy) * sz;
    q[2] = (cx * sy) * cz + (sx * cy) * sz;
    q[3] = (cx * cy) * sz - (sx * sy) * cz;
}

/* ---------------------------------------------------------------- stats ------------------ */
GFO_EXPORT int gfo_stats_clear(GfStepStats* s) {
    if (!s) return GF_E_NULL;
    memset(s, 0, sizeof(*s) * GF_STATS_SHARDS); /* the oracle itself only ever uses shard 0 */
    return GF_OK;
}

int gfo_stats_pack(const GfStatsPackArgs* a);

/* ---------------------------------------------------------------- Phase A ---------------- */
/* genesis_env.py:181-205 (episode_length += 1; last_actions <- actions; actions <- new) then
 * position_action_manager.py:402-414 (NaN/Inf scan, a*scale+offset, clamp) or
 * position_within_limits.py:125-126 (clamp_(-1,1) of the manager's copy, a*scale+offset). */
GFO_EXPORT int gfo_action_step(const GfActionArgs* a) {
    if (!a || !a->actions_in || !a->targets || !a->scale || !a->offset) return GF_E_NULL;
    if (a->mode == GF_ACTION_POSITION && (!a->clip_lo || !a->clip_hi)) return GF_E_NULL;
    const int64_t N = a->num_envs, D = a->num_dofs;
    if (N < 0 || D <= 0) return GF_E_RANGE;
    int flags = 0;
    if (a->episode_length)
        for (int64_t n = 0; n < N; ++n) a->episode_length[n] += 1;
    for (int64_t n = 0; n < N; ++n) {
        for (int64_t d = 0; d < D; ++d) {
            const int64_t k = n * D + d;
            float x = a->actions_in[k];
            if (a->env_actions) {
                a->env_last_actions[k] = a->env_actions[k];
                a->env_actions[k] = x;
            }
            if (a->check_finite && a->mode == GF_ACTION_POSITION) {
                if (isnan(x)) flags |= 1;
                if (isinf(x)) flags |= 2;
            }
            float t;
            if (a->mode == GF_ACTION_WITHIN_LIMITS) {
                /* torch.clamp_(-1, 1): NaN propagates */
                if (x < -1.0f) x = -1.0f;
                if (x > 1.0f) x = 1.0f; /* clamps the manager's private copy, the caller's tensor is untouched */
                t = x * a->scale[d] + a->offset[d];
            } else {
                t = x * a->scale[d] + a->offset[d];
                /* torch.clamp(min=lo, max=hi): min then max, NaN propagates */
                if (t < a->clip_lo[d]) t = a->clip_lo[d];
                if (t > a->clip_hi[d]) t = a->clip_hi[d];
            }
            a->targets[k] = t;
        }
    }
    if (a->stats) a->stats->action_flags |= flags;
    if (a->stats_zero) memset(a->stats_zero, 0, sizeof(GfStepStats) * GF_STATS_SHARDS);
    if (a->stats_fold_src && a->stats_fold_dst) {
        GfStatsPackArgs pk = {a->stats_fold_src, a->stats_fold_dst};
        gfo_stats_pack(&pk);
        if (a->stats_last_reset && a->stats_fold_dst[GF_MAX_TERM_TERMS] > 0.0)
            memcpy(a->stats_last_reset, a->stats_fold_dst, sizeof(double) * GF_STATS_VECTOR_LEN);
    }
    return GF_OK;
}

/* ---------------------------------------------------------------- Phase B2 --------------- */
/* contact/kernel.py:35-90 + contact_manager.py:399-403 (sanitise) + :434-477 (air time). */
GFO_EXPORT int gfo_contact_step(const GfContactArgs* a) {
    if (!a || !a->contacts) return GF_E_NULL;
    const int64_t N = a->num_envs, C = a->num_contacts, L = a->num_targets, W = a->num_with;
    if (L <= 0 || L > GF_MAX_LINK_IDS || W < 0 || W > GF_MAX_LINK_IDS || C < 0) return GF_E_RANGE;
    if (C > 0 && (!a->force || !a->position || !a->link_a || !a->link_b || !a->links_quat)) return GF_E_NULL;
    if (a->track_air_time &&
        (!a->last_air_time || !a->current_air_time || !a->last_contact_time || !a->current_contact_time))
        return GF_E_NULL;
    int flags = 0;
    for (int64_t n = 0; n < N; ++n) {
        float* out_f = a->contacts + n * L * 3;
        float* out_p = a->contact_positions ? a->contact_positions + n * L * 3 : 0;
        float* out_c = a->position_counts ? a->position_counts + n * L : 0;
        for (int64_t t = 0; t < L; ++t) {
            float f[3] = {0, 0, 0}, p[3] = {0, 0, 0}, cnt = 0.0f;
            const int32_t target = a->target_link_ids[t];
            for (int64_t c = 0; c < C; ++c) {
                const int32_t la = a->link_a[n * C + c], lb = a->link_b[n * C + c];
                const int is_a = la == target, is_b = lb == target;
                if (!(is_a || is_b)) continue;
                int include = 1;
                if (a->has_with_filter) {
                    include = 0;
                    for (int64_t w = 0; w < W; ++w) {
                        const int32_t wl = a->with_link_ids[w];
                        if ((is_a && lb == wl) || (is_b && la == wl)) { include = 1; break; }
                    }
                }
                if (!include) continue;
                float fv[3];
                for (int j = 0; j < 3; ++j) {
                    float x = a->force[(n * C + c) * 3 + j];
                    if (isnan(x) || isinf(x)) { x = 0.0f; flags |= 1; } /* nan_to_num(nan=0,posinf=0,neginf=0) */
                    fv[j] = x;
                    p[j] += a->position[(n * C + c) * 3 + j];
                }
                cnt += 1.0f;
                float r[3];
                if (is_b) {
                    rot_inv(a->links_quat + ((int64_t)n * a->num_scene_links + lb) * 4, fv, r);
                } else {
                    const float nf[3] = {-fv[0], -fv[1], -fv[2]};
                    rot_inv(a->links_quat + ((int64_t)n * a->num_scene_links + la) * 4, nf, r);
                }
                for (int j = 0; j < 3; ++j) f[j] += r[j];
            }
            for (int j = 0; j < 3; ++j) {
                out_f[t * 3 + j] = f[j];
                if (out_p) out_p[t * 3 + j] = cnt > 0.0f ? p[j] / cnt : p[j];
            }
            if (out_c) out_c[t] = cnt;
            if (a->links_vel && a->link_vel_out)
                for (int j = 0; j < 3; ++j) a->link_vel_out[(n * L + t) * 3 + j] = a->links_vel[((int64_t)n * a->num_scene_links + target) * 3 + j];
            if (a->links_pos && a->link_pos_out)
                for (int j = 0; j < 3; ++j) a->link_pos_out[(n * L + t) * 3 + j] = a->links_pos[((int64_t)n * a->num_scene_links + target) * 3 + j];
            if (a->track_air_time) {
                const int64_t k = n * L + t;
                const float dt = a->dt;
                const int is_contact = norm3(f) > a->air_time_threshold;
                const float cur_air = a->current_air_time[k], cur_con = a->current_contact_time[k];
                const int new_contact = (cur_air > 0.0f) && is_contact;
                const int new_detach = (cur_con > 0.0f) && !is_contact;
                if (new_contact) a->last_air_time[k] = cur_air + dt;
                a->current_air_time[k] = !is_contact ? cur_air + dt : 0.0f;
                if (new_detach) a->last_contact_time[k] = cur_con + dt;
                a->current_contact_time[k] = is_contact ? cur_con + dt : 0.0f;
            }
        }
    }
    if (a->stats) a->stats->contact_flags |= flags;
    return GF_OK;
}

/* ---------------------------------------------------------------- contact predicates ----- */
static inline int contact_count_over(const GfContactView* v, int64_t n, float thr) {
    int cnt = 0;
    for (int l = 0; l < v->num_links; ++l)
        if (norm3(v->contacts + (n * v->num_links + l) * 3) > thr) ++cnt;
    return cnt;
}

/* ---------------------------------------------------------------- Phase B3 --------------- */
static int eval_termination(const GfTerminationArgs* a, const GfTerm* t, int64_t n) {
    switch (t->op) {
        case GF_T_TIMEOUT: /* terminations.py:17-23 */
            return a->max_episode_length ? a->episode_length[n] > a->max_episode_length[n] : 0;
        case GF_T_BAD_ORIENTATION: { /* terminations.py:52-71; p1 = (float)radians(limit), i0 = grace */
            const int in_grace = a->episode_length[n] <= t->i[0];
            float g[3];
            proj_gravity(&a->entity, n, g);
            float m = norm2(g[0], g[1]);
            if (m > 0.99f) m = 0.99f; /* torch.clamp(max=0.99), NaN propagates */
            const float tilt = asinf(m);
            return !in_grace && (tilt > t->p[1]);
        }
        case GF_T_BASE_HEIGHT_BELOW: /* terminations.py:93-99 */
            return a->entity.pos[3 * n + 2] < t->p[0];
        case GF_T_OUT_OF_BOUNDS: { /* terminations.py:121-137 */
            const float x = a->entity.pos[3 * n], y = a->entity.pos[3 * n + 1];
            return (x < t->p[0]) || (x > t->p[1]) || (y < t->p[2]) || (y > t->p[3]);
        }
        case GF_T_HAS_CONTACT: /* terminations.py:153-155 */
            return contact_count_over(&a->contact[t->i[0]], n, t->p[0]) >= t->i[1];
        case GF_T_CONTACT_FORCE: /* terminations.py:172 */
            return contact_count_over(&a->contact[t->i[0]], n, t->p[0]) > 0;
        case GF_T_CONTACT_FORCE_GRACE: { /* terminations.py:196-205 */
            const int in_grace = a->episode_length[n] <= t->i[1];
            return !in_grace && contact_count_over(&a->contact[t->i[0]], n, t->p[0]) > 0;
        }
        case GF_T_EXTERNAL:
            return a->ext[t->i[0]][n] != 0;
        default:
            return 0;
    }
}

static int check_term_table(const GfTerm* terms, int n, int is_reward) {
    for (int k = 0; k < n; ++k) {
        const int op = terms[k].op;
        if (is_reward ? (op < GF_R_IS_ALIVE || op > GF_R_FOOT_HEIGHT) : (op < GF_T_TIMEOUT || op > GF_T_EXTERNAL))
            return GF_E_OPCODE;
    }
    return GF_OK;
}

/* termination_manager.py:151-190 */
/* The boundary's refusals (include/gf_step.h:44-51): a term that addresses an unbound view, an input a term needs and the
 * descriptor does not carry.  Same codes, same precedence as the library (tests/test_error_codes.py). */
static int check_termination_inputs(const GfTerminationArgs* a) {
    int need_quat = 0, need_pos = 0, need_eplen = 0;
    for (int k = 0; k < a->num_terms; ++k) {
        const GfTerm* t = &a->terms[k];
        switch (t->op) {
            case GF_T_TIMEOUT: if (a->max_episode_length) need_eplen = 1; break;
            case GF_T_BAD_ORIENTATION: need_quat = need_eplen = 1; break;
            case GF_T_BASE_HEIGHT_BELOW:
            case GF_T_OUT_OF_BOUNDS: need_pos = 1; break;
            case GF_T_CONTACT_FORCE_GRACE: need_eplen = 1; /* fallthrough */
            case GF_T_HAS_CONTACT:
            case GF_T_CONTACT_FORCE:
                if (t->i[0] < 0 || t->i[0] >= GF_MAX_CONTACT_VIEWS || !a->contact[t->i[0]].contacts) return GF_E_SLOT;
                if (a->contact[t->i[0]].num_links <= 0) return GF_E_RANGE;
                break;
            case GF_T_EXTERNAL:
                if (t->i[0] < 0 || t->i[0] >= GF_MAX_EXT || !a->ext[t->i[0]]) return GF_E_SLOT;
                break;
            default: return GF_E_OPCODE;
        }
    }
    if (need_quat && !a->entity.quat) return GF_E_NULL;
    if (need_pos && !a->entity.pos) return GF_E_NULL;
    if (need_eplen && !a->episode_length) return GF_E_NULL;
    return GF_OK;
}

GFO_EXPORT int gfo_termination_step(const GfTerminationArgs* a) {
    if (!a || !a->terminated || !a->truncated) return GF_E_NULL;
    if (a->num_terms < 0 || a->num_terms > GF_MAX_TERM_TERMS) return GF_E_RANGE;
    int rc = check_term_table(a->terms, a->num_terms, 0);
    if (rc) return rc;
    rc = check_termination_inputs(a);
    if (rc) return rc;
    const int64_t N = a->num_envs;
    for (int64_t n = 0; n < N; ++n) {
        int term = 0, trunc = 0;
        for (int k = 0; k < a->num_terms; ++k) {
            const int v = eval_termination(a, &a->terms[k], n);
            if (a->terms[k].flags & GF_TERM_FLAG_TIME_OUT) trunc |= v; else term |= v;
            if (v && a->stats) a->stats->term_fired[k] += 1;
            if (a->term_out) a->term_out[(int64_t)k * N + n] = (uint8_t)v;
        }
        a->terminated[n] = (uint8_t)term;
        a->truncated[n] = (uint8_t)trunc;
    }
    return GF_OK;
}

/* ---------------------------------------------------------------- Phase B4 --------------- */
static float eval_reward(const GfRewardArgs* a, const GfTerm* t, int64_t n) {
    const int64_t D = a->num_dofs;
    switch (t->op) {
        case GF_R_IS_ALIVE: return a->terminated[n] ? 0.0f : 1.0f;   /* rewards.py:36-37 */
        case GF_R_TERMINATED: return a->terminated[n] ? 1.0f : 0.0f; /* rewards.py:45-46 */
        case GF_R_BASE_HEIGHT: { /* rewards.py:77-90 */
            float h = a->entity.pos[3 * n + 2];
            if (t->flags & GF_RW_FLAG_TERRAIN) /* rewards.py:84-88 */
                h = h - terrain_height(&a->terrain, a->entity.pos[3 * n], a->entity.pos[3 * n + 1]);
            const float target = (t->flags & GF_RW_FLAG_CMD)
                                     ? a->command[t->i[0]].command[(int64_t)n * cmd_stride(&a->command[t->i[0]])]
                                     : t->p[0];
            const float e = h - target;
            return e * e;
        }
        case GF_R_DOF_SIMILAR_TO_DEFAULT: { /* rewards.py:107-109 */
            float s = 0.0f;
            for (int64_t d = 0; d < D; ++d) s += fabsf(a->dof_pos[n * D + d] - a->default_dof_pos[d]);
            return s;
        }
        case GF_R_LIN_VEL_Z_L2: { /* rewards.py:129-135 */
            float v[3];
            body_lin_vel(&a->entity, n, v);
            return v[2] * v[2];
        }
        case GF_R_ANG_VEL_XY_L2: { /* rewards.py:155-161 */
            float v[3];
            body_ang_vel(&a->entity, n, v);
            return v[0] * v[0] + v[1] * v[1];
        }
        case GF_R_FLAT_ORIENTATION_L2: { /* rewards.py:184-193 */
            float g[3];
            proj_gravity(&a->entity, n, g);
            return g[0] * g[0] + g[1] * g[1];
        }
        case GF_R_BODY_ACCEL_EXP: { /* rewards.py:219-249; state = prev (lin, ang) body-frame velocity */
            float lv[3], av[3], la[3], aa[3];
            body_lin_vel(&a->entity, n, lv);
            body_ang_vel(&a->entity, n, av);
            float* st = a->state[t->i[0]] + n * 6;
            for (int j = 0; j < 3; ++j) {
                if (t->flags & GF_RW_FLAG_FIRST_CALL) {
                    la[j] = 0.0f; aa[j] = 0.0f;
                } else {
                    la[j] = (lv[j] - st[j]) / a->dt;
                    aa[j] = (av[j] - st[3 + j]) / a->dt;
                }
                st[j] = lv[j];
                st[3 + j] = av[j];
            }
            const float motion = norm3(la) + norm3(aa);
            return 1.0f - expf((-t->p[0]) * motion);
        }
        case GF_R_ACTION_RATE_L2: { /* rewards.py:267-271 */
            float s = 0.0f;
            for (int64_t d = 0; d < D; ++d) {
                const float e = a->last_actions[n * D + d] - a->actions[n * D + d];
                s += e * e;
            }
            return s;
        }
        case GF_R_CMD_TRACK_LIN_VEL: { /* rewards.py:304-317 */
            float v[3];
            body_lin_vel(&a->entity, n, v);
            const GfCommandView* c = &a->command[t->i[0]];
            const float e0 = c->command[(int64_t)n * cmd_stride(c)] - v[0];
            const float e1 = c->command[(int64_t)n * cmd_stride(c) + 1] - v[1];
            const float err = e0 * e0 + e1 * e1;
            return expf((-err) / t->p[0]);
        }
        case GF_R_CMD_TRACK_ANG_VEL: { /* rewards.py:345-358 */
            float v[3];
            body_ang_vel(&a->entity, n, v);
            const GfCommandView* c = &a->command[t->i[0]];
            const float e = c->command[(int64_t)n * cmd_stride(c) + t->i[1]] - v[2];
            return expf((-(e * e)) / t->p[0]);
        }
        case GF_R_STAND_STILL: { /* rewards.py:379-385 */
            float s = 0.0f;
            for (int64_t d = 0; d < D; ++d) s += fabsf(a->dof_pos[n * D + d] - a->default_dof_pos[d]);
            const GfCommandView* c = &a->command[t->i[0]];
            const float m = norm2(c->command[(int64_t)n * cmd_stride(c)], c->command[(int64_t)n * cmd_stride(c) + 1]);
            return s * ((m < t->p[0]) ? 1.0f : 0.0f);
        }
        case GF_R_HAS_CONTACT: /* rewards.py:408-410 */
            return contact_count_over(&a->contact[t->i[0]], n, t->p[0]) >= t->i[1] ? 1.0f : 0.0f;
        case GF_R_CONTACT_FORCE: { /* rewards.py:427-428 */
            const GfContactView* v = &a->contact[t->i[0]];
            float s = 0.0f;
            for (int l = 0; l < v->num_links; ++l) {
                float viol = norm3(v->contacts + (n * v->num_links + l) * 3) - t->p[0];
                if (viol < 0.0f) viol = 0.0f; /* clip(min=0) */
                s += viol;
            }
            return s;
        }
        case GF_R_FEET_AIR_TIME: { /* rewards.py:455-469; contact_manager.py:224-226 */
            const GfContactView* v = &a->contact[t->i[0]];
            float s = 0.0f;
            for (int l = 0; l < v->num_links; ++l) {
                const float cc = v->current_contact_time[n * v->num_links + l];
                const float made = ((cc > 0.0f) && (cc < t->p[2])) ? 1.0f : 0.0f;
                float air = (v->last_air_time[n * v->num_links + l] - t->p[0]) * made;
                if ((t->flags & GF_RW_FLAG_MAX) && air > t->p[1]) air = t->p[1];
                s += air;
            }
            if (t->i[1] >= 0) {
                const GfCommandView* c = &a->command[t->i[1]];
                const float m = norm2(c->command[(int64_t)n * cmd_stride(c)], c->command[(int64_t)n * cmd_stride(c) + 1]);
                s = s * ((m > 0.1f) ? 1.0f : 0.0f);
            }
            return s;
        }
        case GF_R_FEET_SLIDE: { /* rewards.py:496-504 */
            const GfContactView* v = &a->contact[t->i[0]];
            float s = 0.0f;
            for (int l = 0; l < v->num_links; ++l) {
                const float c = (norm3(v->contacts + (n * v->num_links + l) * 3) > 1.0f) ? 1.0f : 0.0f;
                s += norm3(v->link_vel + (n * v->num_links + l) * 3) * c;
            }
            return s;
        }
        case GF_R_EXTERNAL: return a->ext[t->i[0]][n];
        case GF_R_GAIT_PHASE: { /* examples/gait_trainer/gait_command_manager.py:295-345 */
            const GfContactView* v = &a->contact[t->i[0]];
            const GfCommandView* gv = &a->command[t->i[1]];
            const float* g = gv->command + (int64_t)n * cmd_stride(gv);
            float quad = 0.0f;
            for (int f = 0; f < 4; ++f) {          /* fl + fr + rl + rr  :300-304 */
                const int l = (t->i[2] >> (8 * f)) & 0xff;
                const float force = norm3(v->contacts + (n * v->num_links + l) * 3);  /* :326-328 */
                const float vel = norm3(v->link_vel + (n * v->num_links + l) * 3);    /* :329 */
                int fl = gait_foot_flags(g[GF_GAIT_PHASE], g[GF_GAIT_OFFSET + f], t->p[1], t->p[2]); /* :332-338 */
                /* `mask.nonzero().flatten()` on the [N,1] masks (:336,338) yields (row, col) pairs, so index 0 (the column)
                 * is in the swing list whenever any env is in swing and in the stance list whenever any env is in stance;
                 * the four assignments :340-343 then leave env 0 with the stance weights, else the swing weights */
                if (n == 0 && a->gait_wave_flags) {
                    unsigned any = 0;
                    for (int64_t b = 0; b < ((int64_t)a->num_envs + 63) / 64; ++b) any |= a->gait_wave_flags[b];
                    if ((any >> (2 * f + 1)) & 1u) fl = 2;
                    else if ((any >> (2 * f)) & 1u) fl = 1;
                }
                const float fw = (fl & 1) ? -1.0f : 0.0f, vw = (fl & 2) ? -1.0f : 0.0f;    /* :340-343 */
                const float foot = vw * vel + fw * force;                             /* :345 */
                quad = f == 0 ? foot : quad + foot;
            }
            return expf(quad);
        }
        case GF_R_FOOT_HEIGHT: { /* :278-293 */
            const GfContactView* v = &a->contact[t->i[0]];
            const GfCommandView* gv = &a->command[t->i[1]];
            const float target = gv->command[(int64_t)n * cmd_stride(gv) + GF_GAIT_HEIGHT];
            float err = 0.0f;
            for (int f = 0; f < 4; ++f) {
                const int l = (t->i[2] >> (8 * f)) & 0xff;
                const float* lv = v->link_vel + (n * v->num_links + l) * 3;
                const float d = v->link_pos[(n * v->num_links + l) * 3 + 2] - target;
                const float e = norm2(lv[0], lv[1]) * (d * d);
                err = f == 0 ? e : err + e;
            }
            return expf((-err) / t->p[0]);
        }
        default: return 0.0f;
    }
}

/* reward_manager.py:166-195 */
static int reward_need_cmd(const GfRewardArgs* a, int idx, int min_width) {
    if (idx < 0 || idx >= GF_MAX_COMMAND_VIEWS || !a->command[idx].command) return GF_E_SLOT;
    return a->command[idx].width >= min_width ? GF_OK : GF_E_RANGE;
}
static int reward_need_contact(const GfRewardArgs* a, int idx) {
    if (idx < 0 || idx >= GF_MAX_CONTACT_VIEWS || !a->contact[idx].contacts) return GF_E_SLOT;
    return a->contact[idx].num_links > 0 ? GF_OK : GF_E_RANGE;
}
static int check_reward_inputs(const GfRewardArgs* a) {
    int quat = 0, pos = 0, lin = 0, ang = 0, term = 0, dofs = 0, acts = 0;
    for (int k = 0; k < a->num_terms; ++k) {
        const GfTerm* t = &a->terms[k];
        int rc = GF_OK;
        if (t->row < 0 || t->row >= GF_MAX_TERMS) return GF_E_RANGE;
        switch (t->op) {
            case GF_R_IS_ALIVE:
            case GF_R_TERMINATED: term = 1; break;
            case GF_R_BASE_HEIGHT:
                pos = 1;
                if (t->flags & GF_RW_FLAG_CMD) rc = reward_need_cmd(a, t->i[0], 1);
                if ((t->flags & GF_RW_FLAG_TERRAIN) && a->terrain.height_field && (a->terrain.rows < 1 || a->terrain.cols < 1)) rc = GF_E_RANGE;
                break;
            case GF_R_DOF_SIMILAR_TO_DEFAULT: dofs = 1; break;
            case GF_R_LIN_VEL_Z_L2: quat = lin = 1; break;
            case GF_R_ANG_VEL_XY_L2: quat = ang = 1; break;
            case GF_R_FLAT_ORIENTATION_L2: quat = 1; break;
            case GF_R_BODY_ACCEL_EXP:
                quat = lin = ang = 1;
                if (t->i[0] < 0 || t->i[0] >= 4 || !a->state[t->i[0]]) rc = GF_E_SLOT;
                break;
            case GF_R_ACTION_RATE_L2: acts = 1; break;
            case GF_R_CMD_TRACK_LIN_VEL: quat = lin = 1; rc = reward_need_cmd(a, t->i[0], 2); break;
            case GF_R_CMD_TRACK_ANG_VEL: quat = ang = 1; rc = reward_need_cmd(a, t->i[0], t->i[1] + 1); if (t->i[1] < 0) rc = GF_E_RANGE; break;
            case GF_R_STAND_STILL: dofs = 1; rc = reward_need_cmd(a, t->i[0], 2); break;
            case GF_R_HAS_CONTACT:
            case GF_R_CONTACT_FORCE: rc = reward_need_contact(a, t->i[0]); break;
            case GF_R_FEET_AIR_TIME:
                rc = reward_need_contact(a, t->i[0]);
                if (!rc && (!a->contact[t->i[0]].last_air_time || !a->contact[t->i[0]].current_contact_time)) rc = GF_E_SLOT;
                if (!rc && t->i[1] >= 0) rc = reward_need_cmd(a, t->i[1], 2);
                break;
            case GF_R_FEET_SLIDE:
                rc = reward_need_contact(a, t->i[0]);
                if (!rc && !a->contact[t->i[0]].link_vel) rc = GF_E_SLOT;
                break;
            case GF_R_EXTERNAL:
                if (t->i[0] < 0 || t->i[0] >= GF_MAX_EXT || !a->ext[t->i[0]]) rc = GF_E_SLOT;
                break;
            case GF_R_GAIT_PHASE:
            case GF_R_FOOT_HEIGHT:
                rc = reward_need_contact(a, t->i[0]);
                if (!rc) rc = reward_need_cmd(a, t->i[1], GF_GAIT_OBS_WIDTH);
                if (!rc && (a->command[t->i[1]].stride < GF_GAIT_ROW || !a->contact[t->i[0]].link_vel)) rc = GF_E_SLOT;
                if (!rc && t->op == GF_R_FOOT_HEIGHT && !a->contact[t->i[0]].link_pos) rc = GF_E_SLOT;
                for (int f = 0; !rc && f < 4; ++f)
                    if (((t->i[2] >> (8 * f)) & 0xff) >= a->contact[t->i[0]].num_links) rc = GF_E_RANGE;
                break;
            default: return GF_E_OPCODE;
        }
        if (rc) return rc;
    }
    if (quat && !a->entity.quat) return GF_E_NULL;
    if (pos && !a->entity.pos) return GF_E_NULL;
    if (lin && !a->entity.lin_vel) return GF_E_NULL;
    if (ang && !a->entity.ang_vel) return GF_E_NULL;
    if (term && !a->terminated) return GF_E_NULL;
    if (dofs && (!a->dof_pos || !a->default_dof_pos || a->num_dofs <= 0)) return GF_E_NULL;
    if (acts && (!a->actions || !a->last_actions || a->num_dofs <= 0)) return GF_E_NULL;
    return GF_OK;
}

GFO_EXPORT int gfo_reward_step(const GfRewardArgs* a) {
    if (!a) return GF_E_NULL;
    if (a->num_terms < 0 || a->num_terms > GF_MAX_TERMS) return GF_E_RANGE;
    int rc = check_term_table(a->terms, a->num_terms, 1);
    if (rc) return rc;
    rc = check_reward_inputs(a);
    if (rc) return rc;
    const int64_t N = a->num_envs;
    if (a->mode == GF_REWARD_MODE_EVAL) {
        if (!a->term_out) return GF_E_NULL;
        for (int64_t n = 0; n < N; ++n)
            for (int k = 0; k < a->num_terms; ++k)
                a->term_out[(int64_t)a->terms[k].row * N + n] = eval_reward(a, &a->terms[k], n);
        return GF_OK;
    }
    if (!a->reward || !a->episode_seconds) return GF_E_NULL;
    if (a->logging_enabled && !a->episode_sums) return GF_E_NULL;
    for (int64_t n = 0; n < N; ++n) {
        float buf = 0.0f;                       /* self._reward_buf[:] = 0.0       :177 */
        a->episode_seconds[n] += a->dt;         /* self._episode_seconds += dt     :178 */
        for (int k = 0; k < a->num_terms; ++k) { /* zero-weight terms are absent from the table :181-182 */
            const float v = eval_reward(a, &a->terms[k], n) * a->terms[k].w; /* fn(...) * (weight*dt)  :185-186 */
            buf += v;                                                        /* :189 */
            if (a->logging_enabled) a->episode_sums[(int64_t)a->terms[k].row * N + n] += v; /* :192-193 */
        }
        a->reward[n] = buf;
    }
    return GF_OK;
}

/* ---------------------------------------------------------------- Phase B5 --------------- */
/* command_manager.py:152-170 (step/reset) and :290-303 (resample_command) */
GFO_EXPORT int gfo_command_step(const GfCommandArgs* a) {
    if (!a || !a->command) return GF_E_NULL;
    if (a->num_ranges <= 0 || a->num_ranges > GF_MAX_RANGES) return GF_E_RANGE;
    if (a->mode == GF_CMD_STEP && (a->resample_steps <= 0 || !a->episode_length)) return a->episode_length ? GF_E_RANGE : GF_E_NULL;
    if (a->mode == GF_CMD_MASKED && !a->mask) return GF_E_NULL;
    const int64_t N = a->num_envs, R = a->num_ranges;
    int count = 0;
    for (int64_t n = 0; n < N; ++n) {
        int go;
        if (a->mode == GF_CMD_STEP) go = (a->episode_length[n] % a->resample_steps) == 0;
        else if (a->mode == GF_CMD_MASKED) go = a->mask[n] || (a->mask2 && a->mask2[n]);
        else go = 1;
        if (!go) continue;
        ++count;
        for (int64_t i = 0; i < R; ++i) {
            const float u = draw_u(a->draws, n * R + i, a->seed, a->stream, (uint32_t)n + a->env_offset, (uint32_t)i);
            a->command[n * R + i] = uniform_range(u, a->lo[i], a->hi[i]);
        }
    }
    if (a->stats && a->mode == GF_CMD_STEP) a->stats->resample_count += count;
    return GF_OK;
}

/* ---------------------------------------------------------------- Phase B5' -------------- */
/* GaitCommandManager (examples/gait_trainer/gait_command_manager.py): step :222-239 (base step command_manager.py:152-162,
 * _log_metrics :430-441), reset :241-255, resample_command/_set_gait :185-211,347-377, gait selection :379-399. */
GFO_EXPORT int gfo_gait_step(const GfGaitArgs* a) {
    if (!a || !a->state || !a->selected) return GF_E_NULL;
    if (a->num_gaits < 1 || a->num_gaits > GF_MAX_GAITS) return GF_E_RANGE;
    if (a->mode == GF_CMD_STEP && (a->resample_steps <= 0 || !a->episode_length)) return a->episode_length ? GF_E_RANGE : GF_E_NULL;
    if (a->mode == GF_CMD_MASKED && !a->mask) return GF_E_NULL;
    const int64_t N = a->num_envs;
    for (int64_t n = 0; n < N; ++n) {
        float* r = a->state + n * GF_GAIT_ROW;
        int go;
        if (a->mode == GF_CMD_STEP) go = (a->episode_length[n] % a->resample_steps) == 0;
        else if (a->mode == GF_CMD_MASKED) go = a->mask[n] || (a->mask2 && a->mask2[n]);
        else go = 1;
        if (go) {
            const uint32_t genv = (uint32_t)n + a->env_offset;
            const float u0 = draw_u(a->draws, n * 3 + 0, a->seed, a->stream, genv, 0);
            const float u1 = draw_u(a->draws, n * 3 + 1, a->seed, a->stream, genv, 1);
            const float u2 = draw_u(a->draws, n * 3 + 2, a->seed, a->stream, genv, 2);
            int g = 0;
            for (int k = 0; k + 1 < a->num_gaits; ++k) g += (u0 >= a->cum_weight[k]) ? 1 : 0;
            a->selected[n] = g;
            for (int f = 0; f < 4; ++f) r[GF_GAIT_OFFSET + f] = a->gait_offsets[g][f];
            r[GF_GAIT_HEIGHT] = ((a->fixed_clearance_mask >> g) & 1) ? a->clearance_lo : uniform_range(u1, a->clearance_lo, a->clearance_hi);
            r[GF_GAIT_PERIOD] = uniform_range(u2, a->period_lo, a->period_hi);
            if (a->mode != GF_CMD_STEP) {
                for (int j = 0; j < 8; ++j) r[GF_GAIT_CLOCK + j] = 0.0f;
                r[GF_GAIT_TIME] = 0.0f;
                r[GF_GAIT_PHASE] = 0.0f;
            }
        }
        if (a->mode == GF_CMD_STEP) {
            if (a->stats && a->selected[n] >= 0 && a->selected[n] < GF_MAX_GAITS) a->stats->gait_count[a->selected[n]] += 1;
            r[GF_GAIT_TIME] = torch_remainder(r[GF_GAIT_TIME] + a->dt, r[GF_GAIT_PERIOD]);  /* :231 */
            r[GF_GAIT_PHASE] = r[GF_GAIT_TIME] / r[GF_GAIT_PERIOD];                        /* :232 */
            for (int f = 0; f < 4; ++f) {                                                  /* :233-239 */
                const float fp = torch_remainder(r[GF_GAIT_PHASE] + r[GF_GAIT_OFFSET + f], 1.0f);
                sincos_det(a->two_pi * fp, &r[GF_GAIT_CLOCK + f], &r[GF_GAIT_CLOCK + 4 + f]);
            }
        }
    }
    if (a->wave_flags) { /* per block of 64 envs: any env with foot f in swing (bit 2f) / stance (bit 2f+1), current state */
        const float pi = 0.5f * a->two_pi;
        for (int64_t b = 0; b < (N + 63) / 64; ++b) {
            unsigned byte = 0;
            for (int64_t n = b * 64; n < N && n < (b + 1) * 64; ++n)
                for (int f = 0; f < 4; ++f)
                    byte |= (unsigned)gait_foot_flags(a->state[n * GF_GAIT_ROW + GF_GAIT_PHASE], a->state[n * GF_GAIT_ROW + GF_GAIT_OFFSET + f], a->two_pi, pi) << (2 * f);
            a->wave_flags[b] = (uint8_t)byte;
        }
    }
    return GF_OK;
}

/* ---------------------------------------------------------------- Phase R ----------------- */
GFO_EXPORT int gfo_masked_reset(const GfResetArgs* a) {
    if (!a || !a->mask) return GF_E_NULL;
    const int64_t N = a->num_envs, D = a->num_dofs;
    if (a->num_reward_terms < 0 || a->num_reward_terms > GF_MAX_TERMS) return GF_E_RANGE;
    if (a->num_contact < 0 || a->num_contact > GF_MAX_CONTACT_VIEWS) return GF_E_RANGE;
    if (a->env_actions && !a->env_last_actions) return GF_E_NULL;
    if (a->scene_dof_pos && !a->default_dof_pos) return GF_E_NULL;
    int count = 0;
    for (int64_t n = 0; n < N; ++n) {
        if (!(a->mask[n] || (a->mask2 && a->mask2[n]))) continue;
        ++count;
        /* GenesisEnv.reset  genesis_env.py:233-252 */
        if (a->env_actions)
            for (int64_t d = 0; d < D; ++d) { a->env_actions[n * D + d] = 0.0f; a->env_last_actions[n * D + d] = 0.0f; }
        if (a->episode_length) a->episode_length[n] = 0;
        if (a->max_episode_length && a->max_random_scaling > 0.0f) {
            const float u = draw_u(a->len_draws, n, a->seed, a->stream, (uint32_t)n + a->env_offset, 0u);
            const float r = uniform_range(u, -1.0f, 1.0f) * a->max_random_scaling;
            a->max_episode_length[n] = (int32_t)rintf((float)a->base_max_episode_length + r); /* torch.round = half-to-even */
        }
        /* RewardManager.reset  reward_manager.py:202-222 */
        if (a->episode_seconds) {
            if (a->reward_logging && a->episode_sums) {
                const float secs = a->episode_seconds[n];
                for (int t = 0; t < a->num_reward_terms; ++t) {
                    float* v = a->episode_sums + (int64_t)t * N + n;
                    if (a->reward_log_mask & (1u << t)) {
                        const float per_sec = *v / secs;
                        if (a->stats) a->stats->reward_episode_sum[t] += (double)per_sec;
                    }
                    *v = 0.0f;
                }
            }
            a->episode_seconds[n] = 1e-10f;
        }
        /* ContactManager.reset  contact_manager.py:316-329 */
        for (int m = 0; m < a->num_contact; ++m)
            for (int s = 0; s < 4; ++s)
                if (a->air_state[m][s])
                    for (int l = 0; l < a->air_links[m]; ++l) a->air_state[m][s][n * a->air_links[m] + l] = 0.0f;
        /* PositionActionManager.reset  position_action_manager.py:455-464 (scene side, synthetic scene only) */
        if (a->scene_dof_pos) {
            for (int64_t d = 0; d < D; ++d) {
                float p = a->default_dof_pos[d];
                if (a->dof_noise_scale != 0.0f) {
                    const float u = draw_u(a->dof_draws, n * D + d, a->seed, a->stream, (uint32_t)n + a->env_offset, (uint32_t)(4 + d));
                    p = p + uniform_range(u, -1.0f, 1.0f) * a->dof_noise_scale;
                }
                a->scene_dof_pos[n * D + d] = p;
                if (a->scene_dof_vel) a->scene_dof_vel[n * D + d] = 0.0f;
            }
        }
        /* mdp.reset.position  reset.py:102-124 */
        if (a->scene_pos) {
            float p[3] = {a->reset_pos[0], a->reset_pos[1], a->reset_pos[2]};
            float q[4] = {a->reset_quat[0], a->reset_quat[1], a->reset_quat[2], a->reset_quat[3]};
            int set_quat = a->set_quat;
            if (a->spawn_mode) {
                /* mdp.reset.randomize_terrain_position (reset.py:199-226):
                 *   terrain_manager.py:236-247  x = rand*(x_max-x_min)+x_min, y likewise (usable centre area), z = height + offset
                 *   reset.py:172-196            every axis given as a (lo, hi) tuple is uniform_, the others stay 0; xyz_to_quat */
                float u[5] = {0, 0, 0, 0, 0};
                const uint32_t genv = (uint32_t)n + a->env_offset;
                if (a->spawn_draws) {
                    for (int j = 0; j < 5; ++j) u[j] = a->spawn_draws[n * 5 + j];
                } else {
                    uint32_t r[4];
                    philox4x32_10(genv, GF_SPAWN_BLOCK, (uint32_t)a->stream, (uint32_t)(a->stream >> 32), (uint32_t)a->seed, (uint32_t)(a->seed >> 32), r);
                    u[0] = (float)(r[0] >> 8) * 5.9604644775390625e-8f;
                    u[1] = (float)(r[1] >> 8) * 5.9604644775390625e-8f;
                    u[4] = (float)(r[2] >> 8) * 5.9604644775390625e-8f;
                    if (a->spawn_rot_mask & 3) {
                        philox4x32_10(genv, GF_SPAWN_BLOCK + 1u, (uint32_t)a->stream, (uint32_t)(a->stream >> 32), (uint32_t)a->seed, (uint32_t)(a->seed >> 32), r);
                        u[2] = (float)(r[0] >> 8) * 5.9604644775390625e-8f;
                        u[3] = (float)(r[1] >> 8) * 5.9604644775390625e-8f;
                    }
                }
                p[0] = u[0] * a->spawn_x_span + a->spawn_x_min;
                p[1] = u[1] * a->spawn_y_span + a->spawn_y_min;
                p[2] = terrain_height(&a->terrain, p[0], p[1]) + a->spawn_height_offset;
                const float rx = (a->spawn_rot_mask & 1) ? uniform_range(u[2], a->spawn_rot_lo[0], a->spawn_rot_hi[0]) : 0.0f;
                const float ry = (a->spawn_rot_mask & 2) ? uniform_range(u[3], a->spawn_rot_lo[1], a->spawn_rot_hi[1]) : 0.0f;
                const float rz = (a->spawn_rot_mask & 4) ? uniform_range(u[4], a->spawn_rot_lo[2], a->spawn_rot_hi[2]) : 0.0f;
                xyz_to_quat(rx, ry, rz, q);
                set_quat = a->spawn_set_quat;
            }
            for (int j = 0; j < 3; ++j) a->scene_pos[3 * n + j] = p[j];
            if (set_quat && a->scene_quat) {
                if (a->quat_stash)
                    for (int j = 0; j < 4; ++j) a->quat_stash[4 * n + j] = a->scene_quat[4 * n + j];
                for (int j = 0; j < 4; ++j) a->scene_quat[4 * n + j] = q[j];
            }
            if (a->zero_velocity) {
                if (a->scene_lin_vel) for (int j = 0; j < 3; ++j) a->scene_lin_vel[3 * n + j] = 0.0f;
                if (a->scene_ang_vel) for (int j = 0; j < 3; ++j) a->scene_ang_vel[3 * n + j] = 0.0f;
                if (a->scene_dof_vel) for (int64_t d = 0; d < D; ++d) a->scene_dof_vel[n * D + d] = 0.0f;
            }
        }
    }
    if (a->stats) a->stats->reset_count += count;
    return GF_OK;
}

/* ---------------------------------------------------------------- Phase O ----------------- */
/* observation_manager.py:218-256 */
GFO_EXPORT int gfo_observe(const GfObservationArgs* a) {
    if (!a || !a->obs) return GF_E_NULL;
    if (a->num_items <= 0 || a->num_items > GF_MAX_OBS_ITEMS) return GF_E_RANGE;
    if (a->ring_slots && (!a->history_ring || (int64_t)a->ring_slots < a->history_len)) return GF_E_RANGE;
    if (a->history_ring < 0 || (int64_t)a->history_ring > (a->ring_slots ? (int64_t)a->ring_slots : (int64_t)a->history_len)) return GF_E_RANGE;
    if (a->history_len < 1 || (a->history_len > 1 && !a->history_ring && !a->prev_obs)) return a->history_len < 1 ? GF_E_RANGE : GF_E_NULL;
    const int64_t N = a->num_envs, D = a->num_dofs, O = a->obs_width, H = a->history_len;
    const int64_t env_stride = (a->history_ring && a->ring_slots) ? O * (int64_t)a->ring_slots : O * H;   /* a ring with more slots than frames (gf_step.h) */
    const int64_t frame_off = a->history_ring ? (int64_t)(a->history_ring - 1) * O : 0;   /* in-place ring: only this slot is written */
    if (O <= 0 || O >= GF_MAX_OBS_WIDTH) return GF_E_RANGE;
    int64_t wsum = 0;
    int need_quat = 0, need_lin = 0, need_ang = 0;
    for (int i = 0; i < a->num_items; ++i) {  /* the boundary's refusals, item by item (tests/test_error_codes.py) */
        const GfObsItem* it = &a->items[i];
        const float* src = NULL;
        int stride = 0;
        if (it->width <= 0) return GF_E_RANGE;
        switch (it->op) {
            case GF_O_COMMAND:
                if (it->i0 < 0 || it->i0 >= GF_MAX_COMMAND_VIEWS || !a->command[it->i0].command) return GF_E_SLOT;
                if (it->width != a->command[it->i0].width) return GF_E_RANGE;
                src = a->command[it->i0].command; stride = cmd_stride(&a->command[it->i0]);
                break;
            case GF_O_DOF_POS: src = a->dof_pos; stride = (int)D; break;
            case GF_O_DOF_VEL: src = a->dof_vel; stride = (int)D; break;
            case GF_O_DOF_FORCE: src = a->dof_force; stride = (int)D; break;
            case GF_O_ACTIONS: src = a->targets; stride = (int)D; break;
            case GF_O_RAW_ACTIONS: src = a->env_actions; stride = (int)D; break;
            case GF_O_EXTERNAL:
                if (it->i0 < 0 || it->i0 >= GF_MAX_EXT || !a->ext[it->i0]) return GF_E_SLOT;
                break;
            case GF_O_BASE_POS:
                if (!a->entity.pos) return GF_E_NULL;
                if (it->width != 3) return GF_E_RANGE;
                break;
            case GF_O_BASE_QUAT:
                if (it->width != 4) return GF_E_RANGE;
                src = a->entity.quat; stride = 4;
                break;
            case GF_O_ANG_VEL_BODY: need_quat = need_ang = 1; if (it->width != 3) return GF_E_RANGE; break;
            case GF_O_LIN_VEL_BODY: need_quat = need_lin = 1; if (it->width != 3) return GF_E_RANGE; break;
            case GF_O_PROJ_GRAVITY: need_quat = 1; if (it->width != 3) return GF_E_RANGE; break;
            case GF_O_CONTACT_FORCE_NORM:
                if (it->i0 < 0 || it->i0 >= GF_MAX_CONTACT_VIEWS || !a->contact[it->i0].contacts) return GF_E_SLOT;
                if (it->width != a->contact[it->i0].num_links) return GF_E_RANGE;
                break;
            default: return GF_E_OPCODE;
        }
        if (stride) {
            if (!src) return GF_E_NULL;
            if (it->width > stride) return GF_E_RANGE;
        }
        wsum += it->width;
    }
    if (wsum != O) return GF_E_RANGE;
    if (need_quat && !a->entity.quat) return GF_E_NULL;
    if (need_lin && !a->entity.lin_vel) return GF_E_NULL;
    if (need_ang && !a->entity.ang_vel) return GF_E_NULL;
    for (int64_t n = 0; n < N; ++n) {
        float* row = a->obs + n * env_stride + frame_off;
        int64_t col = 0;
        /* entity_manager.py:189-195: quaternion cached before the reset of this tick */
        GfEntityView ent = a->entity;
        const int stale = a->stale_quat && ((a->stale_mask && a->stale_mask[n]) || (a->stale_mask2 && a->stale_mask2[n]));
        if (stale) ent.quat = a->stale_quat;
        for (int i = 0; i < a->num_items; ++i) {
            const GfObsItem* it = &a->items[i];
            float tmp[GF_MAX_OBS_WIDTH];
            switch (it->op) {
                case GF_O_COMMAND: {
                    const GfCommandView* c = &a->command[it->i0];
                    for (int j = 0; j < it->width; ++j) tmp[j] = c->command[(int64_t)n * cmd_stride(c) + j];
                } break;
                case GF_O_ANG_VEL_BODY: body_ang_vel(&ent, n, tmp); break;
                case GF_O_LIN_VEL_BODY: body_lin_vel(&ent, n, tmp); break;
                case GF_O_PROJ_GRAVITY: proj_gravity(&ent, n, tmp); break;
                case GF_O_DOF_POS: for (int j = 0; j < it->width; ++j) tmp[j] = a->dof_pos[n * D + j]; break;
                case GF_O_DOF_VEL: for (int j = 0; j < it->width; ++j) tmp[j] = a->dof_vel[n * D + j]; break;
                case GF_O_DOF_FORCE: for (int j = 0; j < it->width; ++j) tmp[j] = a->dof_force[n * D + j]; break;
                case GF_O_ACTIONS: for (int j = 0; j < it->width; ++j) tmp[j] = a->targets[n * D + j]; break;
                case GF_O_RAW_ACTIONS: for (int j = 0; j < it->width; ++j) tmp[j] = a->env_actions[n * D + j]; break;
                case GF_O_CONTACT_FORCE_NORM: {
                    const GfContactView* v = &a->contact[it->i0];
                    for (int j = 0; j < it->width; ++j) tmp[j] = norm3(v->contacts + (n * v->num_links + j) * 3);
                } break;
                case GF_O_EXTERNAL: for (int j = 0; j < it->width; ++j) tmp[j] = a->ext[it->i0][n * it->width + j]; break;
                case GF_O_BASE_POS: for (int j = 0; j < 3; ++j) tmp[j] = a->entity.pos[3 * n + j]; break;
                case GF_O_BASE_QUAT: for (int j = 0; j < 4; ++j) tmp[j] = a->entity.quat[4 * n + j]; break;
                default: return GF_E_OPCODE;
            }
            for (int j = 0; j < it->width; ++j, ++col) {
                float v = tmp[j];
                if (it->scale != 1.0f) v = v * it->scale;             /* :242-244 */
                if (it->noise != 0.0f) {                               /* :247-250 */
                    const float u = draw_u(a->noise_draws, n * O + col, a->seed, a->stream, (uint32_t)n + a->env_offset, (uint32_t)col);
                    v = v + uniform_range(u, -1.0f, 1.0f) * it->noise;
                }
                row[col] = v;
            }
        }
        /* history: newest first (:223-226); in the in-place ring the older frames already sit in their slots */
        if (!a->history_ring)
            for (int64_t k = O; k < O * H; ++k) row[k] = a->prev_obs[n * O * H + (k - O)];
    }
    return GF_OK;
}

/* ---------------------------------------------------------------- entity helpers ---------- */
GFO_EXPORT int gfo_entity_rotate(const GfRotateArgs* a) {
    if (!a || !a->out || !a->entity.quat) return GF_E_NULL;
    for (int64_t n = 0; n < a->num_envs; ++n) {
        float* o = a->out + 3 * n;
        if (a->what == GF_ROT_PROJ_GRAVITY) proj_gravity(&a->entity, n, o);
        else if (a->what == GF_ROT_LIN_VEL) body_lin_vel(&a->entity, n, o);
        else if (a->what == GF_ROT_ANG_VEL) body_ang_vel(&a->entity, n, o);
        else return GF_E_RANGE;
    }
    return GF_OK;
}

/* ---------------------------------------------------------------- synthetic scene --------- */
GFO_EXPORT int gfo_synth_scene_step(const GfSynthSceneArgs* a) {
    if (!a || !a->pos || !a->quat || !a->lin_vel || !a->ang_vel || !a->dof_pos || !a->dof_vel || !a->targets) return GF_E_NULL;
    const int64_t N = a->num_envs, D = a->num_dofs, C = a->num_contacts, NL = a->num_scene_links;
    const float dt = a->dt;
    for (int64_t n = 0; n < N; ++n) {
        /* joints: first-order tracking of the PD target */
        for (int64_t d = 0; d < D; ++d) {
            const float err = a->targets[n * D + d] - a->dof_pos[n * D + d];
            const float v = err * a->joint_rate;
            a->dof_vel[n * D + d] = v;
            a->dof_pos[n * D + d] = a->dof_pos[n * D + d] + v * dt;
        }
        float s[6];
        for (int j = 0; j < 6; ++j) s[j] = philox_uniform(a->seed, a->tick, (uint32_t)n + a->env_offset, (uint32_t)j) * 2.0f - 1.0f;
        float* w = a->ang_vel + 3 * n;
        float* v = a->lin_vel + 3 * n;
        float* p = a->pos + 3 * n;
        float* q = a->quat + 4 * n;
        for (int j = 0; j < 3; ++j) w[j] = w[j] * 0.9f + s[j] * a->ang_noise;
        v[0] = v[0] * 0.9f + s[3] * a->lin_noise;
        v[1] = v[1] * 0.9f + s[4] * a->lin_noise;
        v[2] = (v[2] * 0.9f + (a->height_target - p[2]) * 2.0f) + s[5] * a->lin_noise;
        for (int j = 0; j < 3; ++j) p[j] = p[j] + v[j] * dt;
        /* dq/dt = 0.5 * (0,w) (x) q  (world-frame angular velocity) */
        const float h = 0.5f * dt;
        const float qw = q[0], qx = q[1], qy = q[2], qz = q[3];
        const float dw = ((-(w[0] * qx)) - w[1] * qy) - w[2] * qz;
        const float dx = (w[0] * qw + w[1] * qz) - w[2] * qy;
        const float dy = (w[1] * qw + w[2] * qx) - w[0] * qz;
        const float dz = (w[2] * qw + w[0] * qy) - w[1] * qx;
        float nq[4] = {qw + dw * h, qx + dx * h, qy + dy * h, qz + dz * h};
        const float nrm = sqrtf(((nq[0] * nq[0] + nq[1] * nq[1]) + nq[2] * nq[2]) + nq[3] * nq[3]);
        for (int j = 0; j < 4; ++j) q[j] = nq[j] / nrm;
        if (a->links_quat_out)
            for (int64_t l = 0; l < NL; ++l)
                for (int j = 0; j < 4; ++j) a->links_quat_out[(n * NL + l) * 4 + j] = q[j];
        if (a->links_vel_out)
            for (int64_t l = 0; l < NL; ++l)
                for (int j = 0; j < 3; ++j)
                    a->links_vel_out[(n * NL + l) * 3 + j] = v[j] + (float)(l % 3 == (int64_t)j ? 1 : 0) * 0.05f * w[j];
        if (a->links_pos_out)
            for (int64_t l = 0; l < NL; ++l) {
                /* scene link 0 = ground, 1 = robot base, then (hip, thigh, calf, foot) chains under the body corners */
                float ox = 0.0f, oy = 0.0f, oz = 0.0f;
                if (l > 1) {
                    const int64_t leg = (l - 2) / 4, depth = (l - 2) % 4 + 1;
                    ox = leg < 2 ? 0.19f : -0.19f;
                    oy = (leg & 1) ? -0.11f : 0.11f;
                    oz = -0.085f * (float)depth;
                }
                float* lp = a->links_pos_out + (n * NL + l) * 3;
                lp[0] = p[0] + ox;
                lp[1] = p[1] + oy;
                lp[2] = (p[2] + oz) + 0.03f * w[l % 3];
            }
        if (C > 0 && a->contact_force_out) {
            for (int64_t c = 0; c < C; ++c) {
                const uint32_t col = (uint32_t)(8 + 8 * c);
                const float u_act = philox_uniform(a->seed, a->tick, (uint32_t)n + a->env_offset, col);
                const float u_link = philox_uniform(a->seed, a->tick, (uint32_t)n + a->env_offset, col + 1);
                const float fx = philox_uniform(a->seed, a->tick, (uint32_t)n + a->env_offset, col + 2) * 2.0f - 1.0f;
                const float fy = philox_uniform(a->seed, a->tick, (uint32_t)n + a->env_offset, col + 3) * 2.0f - 1.0f;
                float fz = philox_uniform(a->seed, a->tick, (uint32_t)n + a->env_offset, col + 4);
                int active = u_act < a->contact_prob;
                const int64_t k = n * C + c;
                int32_t lb = 1 + (int32_t)(u_link * (float)(NL - 1));
                int robot_on_a = 0;
                if (a->foot_link_mask) {   /* walking model (gf_step.h, GfSynthSceneArgs.foot_link_mask) */
                    const uint32_t feet = a->foot_link_mask, genv = (uint32_t)n + a->env_offset;
                    int n_feet = 0;
                    for (uint32_t m = feet; m; m &= m - 1u) ++n_feet;
                    if (c < n_feet) {
                        uint32_t m = feet;
                        for (int64_t j = 0; j < c; ++j) m &= m - 1u;
                        lb = 0;
                        while (!((m >> lb) & 1u)) ++lb;
                        const uint32_t ph = ((uint32_t)a->tick + genv * 7u) % 20u;
                        const int pair_a = (((int)c ^ ((int)c >> 1)) & 1) == 0;
                        const int stance = pair_a ? ph < 10u : ph >= 10u;
                        const float pr = stance ? fminf(1.8f * a->foot_contact_prob, 1.0f) : 0.2f * a->foot_contact_prob;
                        active = u_act < pr;
                        robot_on_a = u_link < 0.5f;
                        fz = robot_on_a ? u_link * 2.0f : u_link * 2.0f - 1.0f;   /* (the side's draw, stretched back onto [0, 1): exact) */
                    } else {
                        const int32_t kk = (int32_t)(u_link * (float)(2 * (NL - 1)));
                        lb = 1 + (kk >> 1);
                        robot_on_a = (kk & 1) != 0;
                    }
                }
                if (lb > (int32_t)NL - 1) lb = (int32_t)NL - 1;
                const float sx = fx * a->contact_force * 0.25f, sy = fy * a->contact_force * 0.25f, sz = fz * a->contact_force;
                a->link_a_out[k] = active ? (robot_on_a ? lb : 0) : -1;
                a->link_b_out[k] = active ? (robot_on_a ? 0 : lb) : -1;
                a->contact_force_out[k * 3 + 0] = active ? (robot_on_a ? -sx : sx) : 0.0f;
                a->contact_force_out[k * 3 + 1] = active ? (robot_on_a ? -sy : sy) : 0.0f;
                a->contact_force_out[k * 3 + 2] = active ? (robot_on_a ? -sz : sz) : 0.0f;
                a->contact_pos_out[k * 3 + 0] = active ? p[0] + fx * 0.2f : 0.0f;
                a->contact_pos_out[k * 3 + 1] = active ? p[1] + fy * 0.2f : 0.0f;
                a->contact_pos_out[k * 3 + 2] = 0.0f;
            }
        }
    }
    return GF_OK;
}

/* Rollout-storage write: rsl_rl's `observations[t+1].copy_(obs)`, `rewards[t].copy_(rew)`,
 * `dones[t].copy_(terminated | truncated)` (call site examples/simple/train.py:125-129). */
GFO_EXPORT int gfo_rollout_write(const GfRolloutArgs* a) {
    if (!a) return GF_E_NULL;
    if (a->num_envs < 0 || a->obs_width < 0) return GF_E_RANGE;
    if ((a->obs_out && !a->obs) || (a->reward_out && !a->reward) || (a->done_out && !a->terminated)) return GF_E_NULL;
    const int64_t N = a->num_envs;
    if (a->obs_out) memcpy(a->obs_out, a->obs, (size_t)N * (size_t)a->obs_width * sizeof(float));
    if (a->reward_out) memcpy(a->reward_out, a->reward, (size_t)N * sizeof(float));
    if (a->done_out)
        for (int64_t n = 0; n < N; ++n) a->done_out[n] = (uint8_t)((a->terminated[n] != 0) | (a->truncated && a->truncated[n] != 0));
    return GF_OK;
}

/* The policy's rows of a transition + time-out bootstrap (rsl_rl RolloutStorage.add_transitions / PPO.process_env_step — a
 * third-party package, not under /root/reference: configured and called at examples/simple/train.py:37-79,125-129; the
 * algorithm is restated in include/gf_step.h). */
GFO_EXPORT int gfo_rollout_policy_write(const GfRolloutPolicyArgs* a) {
    if (!a) return GF_E_NULL;
    if (a->num_envs < 0 || a->num_actions < 0) return GF_E_RANGE;
    if ((a->actions_out && !a->actions) || (a->mu_out && !a->mu) || (a->sigma_out && !a->sigma) || (a->values_out && !a->values) ||
        (a->log_prob_out && !a->log_prob) || (a->time_outs && (!a->reward_row || !a->values)))
        return GF_E_NULL;
    const size_t N = (size_t)a->num_envs, row = N * (size_t)a->num_actions * sizeof(float);
    if (a->actions_out) memcpy(a->actions_out, a->actions, row);
    if (a->mu_out) memcpy(a->mu_out, a->mu, row);
    if (a->sigma_out) memcpy(a->sigma_out, a->sigma, row);
    if (a->values_out) memcpy(a->values_out, a->values, N * sizeof(float));
    if (a->log_prob_out) memcpy(a->log_prob_out, a->log_prob, N * sizeof(float));
    if (a->time_outs)
        for (size_t n = 0; n < N; ++n) a->reward_row[n] = a->reward_row[n] + (a->gamma * a->values[n]) * (a->time_outs[n] ? 1.0f : 0.0f);
    return GF_OK;
}

/* The ascending index list of the listed envs: torch's `(terminated | truncated).nonzero()` (managed_env.py:308-310). */
GFO_EXPORT int gfo_done_compact(const GfCompactArgs* a) {
    if (!a || !a->mask || !a->ids_out || !a->count_out || !a->block_counts) return GF_E_NULL;
    if (a->num_envs < 0 || a->num_envs >= ((int64_t)1 << 31)) return GF_E_RANGE;
    int32_t k = 0;
    for (int64_t n = 0; n < a->num_envs; ++n)
        if (a->mask[n] || (a->mask2 && a->mask2[n])) a->ids_out[k++] = n;
    *a->count_out = k;
    return GF_OK;
}

/* GAE (rsl_rl RolloutStorage.compute_returns; gamma / lam from examples/simple/train.py:41-47): the torch loop, scalar. */
GFO_EXPORT int gfo_gae(const GfGaeArgs* a) {
    if (!a || !a->rewards || !a->values || !a->dones || !a->last_values || !a->returns || !a->advantages) return GF_E_NULL;
    if (a->num_envs < 0 || a->num_steps < 1) return GF_E_RANGE;
    if (a->normalize && !a->moments) return GF_E_NULL;
    const int64_t N = a->num_envs;
    const int T = a->num_steps;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t n = 0; n < N; ++n) {
        float next = a->last_values[n], adv = 0.0f;
        for (int t = T - 1; t >= 0; --t) {
            const int64_t at = (int64_t)t * N + n;
            const float nt = 1.0f - (a->dones[at] ? 1.0f : 0.0f);
            const float delta = (a->rewards[at] + (nt * a->gamma) * next) - a->values[at];
            adv = delta + ((nt * a->gamma) * a->lam) * adv;
            const float ret = adv + a->values[at];
            a->returns[at] = ret;
            a->advantages[at] = ret - a->values[at];
            s1 += (double)a->advantages[at];
            s2 += (double)a->advantages[at] * (double)a->advantages[at];
            next = a->values[at];
        }
    }
    if (a->moments) { a->moments[0] = s1; a->moments[1] = s2; }
    if (a->normalize) {
        const int64_t total = N * T;
        const double cnt = (double)total, mean = total ? s1 / cnt : 0.0;
        const double var = total > 1 ? (s2 - s1 * mean) / (cnt - 1.0) : 0.0;
        const float mu = (float)mean, denom = (float)sqrt(var > 0.0 ? var : 0.0) + 1e-8f;
        for (int64_t i = 0; i < total; ++i) a->advantages[i] = (a->advantages[i] - mu) / denom;
    }
    return GF_OK;
}

/* Host twin of gf_post_physics_step: BY DEFINITION the phases in the reference's order (managed_env.py:303-326). */
GFO_EXPORT int gfo_post_physics_check(const GfPostRefs* r) { return (r && r->termination && (r->reset || (r->flags & GF_POST_NO_RESET))) ? GF_OK : GF_E_NULL; }

GFO_EXPORT int gfo_post_physics_step(const GfPostRefs* r) {
    if (!r || !r->termination) return GF_E_NULL;
    if (r->flags & GF_POST_NO_RESET) {   /* the front of a step whose reset goes through user code: nothing behind the step phases */
        int rc0 = GF_OK;
        if (r->num_observe || r->rollout || (r->flags & GF_POST_OBSERVE_ONLY)) return GF_E_UNSUPPORTED;
        if (!(r->flags & GF_POST_TERMINATION_DONE) && (rc0 = gfo_termination_step(r->termination))) return rc0;
        if (r->reward && (rc0 = gfo_reward_step(r->reward))) return rc0;
        for (int c = 0; c < r->num_command; ++c)
            if ((rc0 = gfo_command_step(r->command_step[c]))) return rc0;
        for (int g = 0; g < r->num_gait; ++g)
            if ((rc0 = gfo_gait_step(r->gait_step[g]))) return rc0;
        for (int g = 0; g < r->num_gait; ++g)
            if (r->gait_flags_next[g] && r->gait_step[g]->wave_flags) {
                const int64_t blocks = ((int64_t)r->gait_step[g]->num_envs + 63) / 64;
                memcpy(r->gait_flags_next[g], r->gait_step[g]->wave_flags, (size_t)blocks);
            }
        return GF_OK;
    }
    if (!r->reset) return GF_E_NULL;
    int rc = GF_OK;
    if (r->flags & GF_POST_OBSERVE_ONLY) {   /* everything up to the reset has run as calls of their own: the observations are left */
        if (r->reward || r->num_command || r->num_gait) return GF_E_UNSUPPORTED;
        for (int o = 0; o < r->num_observe; ++o)
            if ((rc = gfo_observe(r->observe[o]))) return rc;
        return GF_OK;
    }
    /* GF_POST_TERMINATION_DONE: the termination phase already ran as a call of its own (Python-level terms between it and the
     * reward phase, managed_env.py:303-319) */
    if (!(r->flags & GF_POST_TERMINATION_DONE) && (rc = gfo_termination_step(r->termination))) return rc;
    if (r->reward && (rc = gfo_reward_step(r->reward))) return rc;
    for (int c = 0; c < r->num_command; ++c)
        if ((rc = gfo_command_step(r->command_step[c]))) return rc;
    for (int g = 0; g < r->num_gait; ++g)
        if ((rc = gfo_gait_step(r->gait_step[g]))) return rc;
    if ((rc = gfo_masked_reset(r->reset))) return rc;
    for (int c = 0; c < r->num_command; ++c)
        if ((rc = gfo_command_step(r->command_reset[c]))) return rc;
    for (int g = 0; g < r->num_gait; ++g)
        if ((rc = gfo_gait_step(r->gait_reset[g]))) return rc;
    for (int o = 0; o < r->num_observe; ++o)
        if ((rc = gfo_observe(r->observe[o]))) return rc;
    /* the single launch writes the swing / stance bytes of the state it leaves into the OTHER buffer (GfPostRefs); in sequence
     * gait.step + gait.reset have just left exactly those bytes in wave_flags */
    if (r->rollout && (rc = gfo_rollout_write(r->rollout))) return rc;
    for (int g = 0; g < r->num_gait; ++g)
        if (r->gait_flags_next[g] && r->gait_step[g]->wave_flags) {
            const int64_t blocks = ((int64_t)r->gait_step[g]->num_envs + 63) / 64;
            memcpy(r->gait_flags_next[g], r->gait_step[g]->wave_flags, (size_t)blocks);
        }
    return GF_OK;
}

GFO_EXPORT int gfo_stats_pack(const GfStatsPackArgs* a) {
    if (!a || !a->src || !a->dst) return GF_E_NULL;
    for (int v = 0; v < GF_STATS_VECTOR_LEN; ++v) {
        double acc = 0.0;
        for (int s = 0; s < GF_STATS_SHARDS; ++s) {
            const GfStepStats* b = &a->src[s];
            double x;
            if (v < GF_MAX_TERM_TERMS) x = (double)b->term_fired[v];
            else if (v == GF_MAX_TERM_TERMS) x = (double)b->reset_count;
            else if (v == GF_MAX_TERM_TERMS + 1) x = (double)(b->action_flags & 1);
            else if (v == GF_MAX_TERM_TERMS + 2) x = (double)((b->action_flags >> 1) & 1);
            else if (v == GF_MAX_TERM_TERMS + 3) x = (double)(b->contact_flags & 1);
            else if (v == GF_MAX_TERM_TERMS + 4) x = (double)b->resample_count;
            else if (v < GF_MAX_TERM_TERMS + 5 + GF_MAX_TERMS) x = b->reward_episode_sum[v - (GF_MAX_TERM_TERMS + 5)];
            else x = (double)b->gait_count[v - (GF_MAX_TERM_TERMS + 5 + GF_MAX_TERMS)];
            const int is_flag = v > GF_MAX_TERM_TERMS && v < GF_MAX_TERM_TERMS + 4;
            acc = is_flag ? (x > acc ? x : acc) : acc + x;
        }
        a->dst[v] = acc;
    }
    return GF_OK;
}

/* observation_manager.py:219-226: the returned tensor is cat(history), newest frame first; with the history kept as a ring
 * (slot k = newest, then upwards, wrapping) that is a gather */
GFO_EXPORT int gfo_history_unroll(const GfHistoryUnrollArgs* a) {
    if (!a) return GF_E_NULL;
    if (a->num_envs < 0 || a->frame_width < 1 || a->history_len < 1 || a->ring_slot < 1 || a->ring_slot > a->history_len) return GF_E_RANGE;
    if ((int64_t)a->frame_width * a->history_len >= (1 << 17)) return GF_E_RANGE;
    if (a->num_envs == 0) return GF_OK;
    if (!a->ring || !a->out) return GF_E_NULL;
    if (((uintptr_t)a->out & 15u) || ((uintptr_t)a->out2 & 15u) || ((uintptr_t)a->ring & 3u)) return GF_E_UNSUPPORTED;
    const int O = a->frame_width, H = a->history_len, k = a->ring_slot - 1;
    for (int64_t n = 0; n < a->num_envs; ++n)
        for (int j = 0; j < H; ++j) {
            const float* src = a->ring + ((size_t)n * H + (size_t)((k + j) % H)) * O;
            memcpy(a->out + ((size_t)n * H + j) * O, src, sizeof(float) * (size_t)O);
            if (a->out2) memcpy(a->out2 + ((size_t)n * H + j) * O, src, sizeof(float) * (size_t)O);
        }
    return GF_OK;
}

GFO_EXPORT int gfo_stats_last_reset(const double* rows, int num_rows, double* dst) {
    if (!rows || !dst) return GF_E_NULL;
    if (num_rows < 0 || num_rows > 64) return GF_E_RANGE;
    for (int r = num_rows - 1; r >= 0; --r)
        if (rows[(size_t)r * GF_STATS_VECTOR_LEN + GF_MAX_TERM_TERMS] > 0.0) {
            memcpy(dst, rows + (size_t)r * GF_STATS_VECTOR_LEN, sizeof(double) * GF_STATS_VECTOR_LEN);
            break;
        }
    return GF_OK;
}

/* host twin of gf_post_physics_step_contacts: by definition the managers' steps, then the post-physics phases, in sequence
 * (managed_env.py:294-326) */
GFO_EXPORT int gfo_post_physics_step_contacts(const GfPostRefs* r, const GfContactArgs* const* contacts, int num_contacts) {
    if (num_contacts < 0 || (num_contacts > 0 && !contacts)) return GF_E_NULL;
    for (int m = 0; m < num_contacts; ++m) {
        const int rc = gfo_contact_step(contacts[m]);
        if (rc != GF_OK) return rc;
    }
    return gfo_post_physics_step(r);
}

/* host twin of gf_run_ops (the recorded-step replay), so the trace/replay host logic is testable on CPU */
GFO_EXPORT int gfo_run_ops(const GfOp* ops, int num_ops, int* failed_index) {
    if (!ops || num_ops < 0) return GF_E_NULL;
    for (int i = 0; i < num_ops; ++i) {
        int rc = GF_OK;
        const void* a = ops[i].args;
        switch (ops[i].phase) {
            case GF_PHASE_ACTION: rc = gfo_action_step((const GfActionArgs*)a); break;
            case GF_PHASE_CONTACT: rc = gfo_contact_step((const GfContactArgs*)a); break;
            case GF_PHASE_TERMINATION: rc = gfo_termination_step((const GfTerminationArgs*)a); break;
            case GF_PHASE_REWARD: rc = gfo_reward_step((const GfRewardArgs*)a); break;
            case GF_PHASE_COMMAND: rc = gfo_command_step((const GfCommandArgs*)a); break;
            case GF_PHASE_RESET: rc = gfo_masked_reset((const GfResetArgs*)a); break;
            case GF_PHASE_OBSERVE: rc = gfo_observe((const GfObservationArgs*)a); break;
            case GF_PHASE_ROTATE: rc = gfo_entity_rotate((const GfRotateArgs*)a); break;
            case GF_PHASE_SCENE: rc = gfo_synth_scene_step((const GfSynthSceneArgs*)a); break;
            case GF_PHASE_TERRAIN: rc = gfo_terrain_height((const GfTerrainHeightArgs*)a); break;
            case GF_PHASE_GAIT: rc = gfo_gait_step((const GfGaitArgs*)a); break;
            case GF_OP_STATS_CLEAR: rc = gfo_stats_clear((GfStepStats*)a); break;
            case GF_OP_POST_PHYSICS: rc = gfo_post_physics_step((const GfPostRefs*)a); break;
            case GF_PHASE_ROLLOUT: rc = gfo_rollout_write((const GfRolloutArgs*)a); break;
            case GF_PHASE_UNROLL: rc = gfo_history_unroll((const GfHistoryUnrollArgs*)a); break;
            case GF_PHASE_ROLLOUT_POLICY: rc = gfo_rollout_policy_write((const GfRolloutPolicyArgs*)a); break;
            case GF_PHASE_GAE: rc = gfo_gae((const GfGaeArgs*)a); break;
            case GF_PHASE_COMPACT: rc = gfo_done_compact((const GfCompactArgs*)a); break;
            case GF_OP_STATS_PACK: rc = gfo_stats_pack((const GfStatsPackArgs*)a); break;
            case GF_OP_STATS_COPY: {
                const GfStatsCopyArgs* c = (const GfStatsCopyArgs*)a;
                if (!c || !c->src || !c->dst) { rc = GF_E_NULL; break; }
                memcpy(c->dst, c->src, sizeof(GfStepStats) * GF_STATS_SHARDS);
            } break;
            default: rc = GF_E_OPCODE; break;
        }
        if (rc != GF_OK) {
            if (failed_index) *failed_index = i;
            return rc;
        }
    }
    return GF_OK;
}


/* the patch table of a recorded step (gf_step.h: GfReplay) */
static int replay_patch(const GfReplay* r, const void* actions, const void* const* params, int num_params) {
    if (!r || (r->num_patches > 0 && !r->patches)) return GF_E_NULL;
    for (int i = 0; i < r->num_patches; ++i) {
        const GfReplayPatch* p = &r->patches[i];
        switch (p->kind) {
            case GF_PATCH_ACTIONS:
                if (!p->target) return GF_E_NULL;
                *(const void**)p->target = actions;
                break;
            case GF_PATCH_STREAM:
                if (!p->target || !r->rng_stream) return GF_E_NULL;
                *(uint64_t*)p->target = ++*r->rng_stream;
                break;
            case GF_PATCH_COUNTER:
                if (!p->target || !p->aux) return GF_E_NULL;
                *(uint64_t*)p->target = (*(uint64_t*)p->aux)++;
                break;
            case GF_PATCH_ROTATE: {
                GfRotor* ro = (GfRotor*)p->aux;
                if (!ro || ro->count < 1 || ro->count > 8 || ro->cur < 0 || ro->cur >= ro->count) return GF_E_RANGE;
                if (p->target) *(void**)p->target = ro->slot[ro->cur];
                ro->cur = (ro->cur + 1) % ro->count;
                if (p->target2) *(void**)p->target2 = ro->slot[ro->cur];
            } break;
            case GF_PATCH_PARAM:
                if (!p->target) return GF_E_NULL;
                if (p->index < 0 || p->index >= num_params || !params) return GF_E_RANGE;
                *(const void**)p->target = params[p->index];
                break;
            case GF_PATCH_PARAM_OFFSET:
                if (!p->target) return GF_E_NULL;
                if (p->index < 0 || p->index >= num_params || !params) return GF_E_RANGE;
                if (!params[p->index]) return GF_E_NULL;
                *(const char**)p->target = (const char*)params[p->index] + (intptr_t)p->aux;
                break;
            case GF_PATCH_COPY:
                if (!p->target || !p->aux) return GF_E_NULL;
                *(uint64_t*)p->target = *(const uint64_t*)p->aux;
                break;
            case GF_PATCH_RING_SLOT: {
                GfRingClock* c = (GfRingClock*)p->aux;
                if (!p->target || !c || c->length < 1) return GF_E_RANGE;
                *(int32_t*)p->target = (c->length - c->calls % c->length) % c->length + 1;
                if (p->target2) *(int32_t*)p->target2 = *(int32_t*)p->target;
                ++c->calls;
            } break;
            default: return GF_E_OPCODE;
        }
    }
    return GF_OK;
}

GFO_EXPORT int gfo_replay_step(const GfReplay* r, const void* actions, const void* const* params, int num_params, int* failed_index) {
    if (failed_index) *failed_index = -1;
    const int rc = replay_patch(r, actions, params, num_params);
    if (rc != GF_OK || !r->ops || r->num_ops <= 0) return rc;
    return gfo_run_ops(r->ops, r->num_ops, failed_index);
}

GFO_EXPORT int gfo_abi_version(void) { return GF_ABI_VERSION; }
GFO_EXPORT int gfo_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(GfStepStats);
        case 1: return (int)sizeof(GfActionArgs);
        case 2: return (int)sizeof(GfContactArgs);
        case 3: return (int)sizeof(GfTerminationArgs);
        case 4: return (int)sizeof(GfRewardArgs);
        case 5: return (int)sizeof(GfCommandArgs);
        case 6: return (int)sizeof(GfResetArgs);
        case 7: return (int)sizeof(GfObservationArgs);
        case 8: return (int)sizeof(GfRotateArgs);
        case 9: return (int)sizeof(GfSynthSceneArgs);
        case 10: return (int)sizeof(GfTerm);
        case 11: return (int)sizeof(GfObsItem);
        case 12: return (int)sizeof(GfTerrainView);
        case 13: return (int)sizeof(GfTerrainHeightArgs);
        case 14: return (int)sizeof(GfGaitArgs);
        case 15: return (int)sizeof(GfContactView);
        case 16: return (int)sizeof(GfCommandView);
        case 17: return (int)sizeof(GfPostRefs);
        case 18: return (int)sizeof(GfRolloutArgs);
        case 19: return (int)sizeof(GfHistoryUnrollArgs);
        case 20: return (int)sizeof(GfRolloutPolicyArgs);
        case 21: return (int)sizeof(GfGaeArgs);
        case 22: return (int)sizeof(GfCompactArgs);
        default: return -1;
    }
}
