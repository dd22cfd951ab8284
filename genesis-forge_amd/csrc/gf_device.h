// gf_device.h — device-side helpers shared by the phase kernels (gfx950 / CDNA4 only).
//
// Arithmetic contract: f32, one rounding per operation (the library is built with
// -ffp-contract=off so hipcc never fuses a*b+c), in the operation order of the reference's torch
// expressions, so masks come out bit-identical to the reference's CPU run and floats differ only
// where a transcendental (expf) is involved.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gf_step.h"

#define GF_WAVE 64

#define GF_HIP_CHECK(expr)                         \
    do {                                           \
        hipError_t _e = (expr);                    \
        if (_e != hipSuccess) return (int)_e;      \
    } while (0)

#define GF_GLOBAL __attribute__((address_space(1)))

namespace gf {

// Zero pad inside the code object.  Kernels redirect the loads of inputs a config does not need
// to it (a broadcast L2 hit) instead of branching around them, which keeps the load stream
// straight-line: no phi copies, no per-block waits, every real load in flight at once.
static __device__ __attribute__((aligned(16))) float g_zero_pad[64];  // one copy per translation unit

// need ? (global)p + off : (global)zero_pad — both selects are wave-uniform (SGPR base, VGPR offset).
template <typename T>
__device__ __forceinline__ const GF_GLOBAL T* gsel(bool need, const T* p, uint32_t elem_off) {
    const GF_GLOBAL T* base = need ? (const GF_GLOBAL T*)p : (const GF_GLOBAL T*)g_zero_pad;
    return base + (need ? elem_off : 0u);
}

// explicit global-memory views of pointers whose address space the compiler cannot infer (descriptor fields read
// back from LDS): keeps loads/stores on the global_* path instead of flat_* (which also ties up lgkmcnt)
template <typename T>
__device__ __forceinline__ GF_GLOBAL T* G(T* p) { return (GF_GLOBAL T*)p; }
template <typename T>
__device__ __forceinline__ const GF_GLOBAL T* G(const T* p) { return (const GF_GLOBAL T*)p; }

// 16-byte global load through an address_space(1) pointer (float4 is a class type and cannot bind
// to a non-generic address space; the ext-vector can).
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4(const GF_GLOBAL float* p) {
    const f32x4 v = *reinterpret_cast<const GF_GLOBAL f32x4*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011).  counter = (env, col/4, stream_lo, stream_hi), key = seed.
// Integer-only, so host oracle, numpy model and this kernel agree bit for bit.
// ---------------------------------------------------------------------------------------------
struct U4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ float u24_to_unit(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }

__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t stream, uint32_t env, uint32_t col) {
    const U4 r = philox4x32_10(env, col >> 2, (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t s = col & 3u;
    const uint32_t v = s == 0 ? r.x : (s == 1 ? r.y : (s == 2 ? r.z : r.w));
    return u24_to_unit(v);
}

__device__ __forceinline__ float draw_u(const float* __restrict__ draws, int64_t idx, uint64_t seed, uint64_t stream, uint32_t env,
                                        uint32_t col) {
    return draws ? draws[idx] : philox_uniform(seed, stream, env, col);
}

// Tensor.uniform_(lo, hi) == u*(hi-lo)+lo  (command_manager.py:302, genesis_env.py:249)
__device__ __forceinline__ float uniform_range(float u, float lo, float hi) { return u * (hi - lo) + lo; }

// ---------------------------------------------------------------------------------------------
// Quaternion helpers: transform_by_quat(v, inv_quat(q))  (utils.py:13-55, entity_manager.py:130-146)
// ---------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};

__device__ __forceinline__ V3 rot_inv(const float4 q, const V3 v) {
    const float w = q.x, a = -q.y, b = -q.z, c = -q.w;  // (w,x,y,z) conjugated
    const float t0 = (b * v.z - c * v.y) * 2.0f;
    const float t1 = (c * v.x - a * v.z) * 2.0f;
    const float t2 = (a * v.y - b * v.x) * 2.0f;
    V3 o;
    o.x = (v.x + w * t0) + (b * t2 - c * t1);
    o.y = (v.y + w * t1) + (c * t0 - a * t2);
    o.z = (v.z + w * t2) + (a * t1 - b * t0);
    return o;
}

__device__ __forceinline__ float norm3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }
__device__ __forceinline__ float norm2(float x, float y) { return sqrtf(x * x + y * y); }

__device__ __forceinline__ V3 load3(const float* __restrict__ p, int64_t n) {
    const float* r = p + 3 * n;
    return V3{r[0], r[1], r[2]};
}
__device__ __forceinline__ float4 load_quat(const float* __restrict__ p, int64_t n) {
    return reinterpret_cast<const float4*>(p)[n];
}

// row stride of a command view in floats (0 = dense)
__device__ __forceinline__ int cmd_stride(const GfCommandView& c) { return c.stride ? c.stride : c.width; }

// torch's `%` on float tensors (aten remainder): fmod, then moved to the divisor's sign
__device__ __forceinline__ float torch_remainder(float a, float b) {
    // fmod is exact by definition, so the two cheap cases give libm's bits: a in [0, b) is its own remainder and a in [b, 2b) leaves
    // a − b (exact, Sterbenz).  A periodic clock advanced by one tick is always one of the two; libm's generic fmodf — a data-dependent
    // loop — runs only for the rest (negative, huge, inf / NaN operands)
    float m;
    if (a >= 0.0f && a < b) m = a;
    else if (a >= b && a < b + b) m = a - b;
    else m = fmodf(a, b);
    if ((m != 0.0f) && ((b < 0.0f) != (m < 0.0f))) m += b;
    return m;
}
// torch.remainder(a, 1): fmodf(a, 1) is the fraction with a's sign, and a − trunc(a) is exact in f32 (inf − inf and NaN give NaN as
// fmodf does; −2 → −0).  Branch-free: behind libm's loop the compiler could issue none of the loads that follow it
__device__ __forceinline__ float torch_remainder_one(float a) {
    float m = copysignf(a - truncf(a), a);
    if (m < 0.0f) m += 1.0f;
    return m;
}

// swing / stance of one foot in the gait cycle (examples/gait_trainer/gait_command_manager.py:331-338): bit 0 = swing
// (0 <= phi < π), bit 1 = stance (π <= phi < 2π), phi = fmod(phase + offset, 1)·(float)2π; NaN is neither
__device__ __forceinline__ int gait_foot_flags(float phase, float offset, float two_pi, float pi) {
    const float phi = torch_remainder_one(phase + offset) * two_pi;
    return (((phi >= 0.0f) && (phi < pi)) ? 1 : 0) | (((phi >= pi) && (phi < two_pi)) ? 2 : 0);
}

// NaN-propagating clamps with torch semantics (Appendix B of SURVEY.md)
__device__ __forceinline__ float clamp_max(float x, float hi) { return x > hi ? hi : x; }
__device__ __forceinline__ float clamp_min(float x, float lo) { return x < lo ? lo : x; }

// ---------------------------------------------------------------------------------------------
// TerrainManager.get_terrain_height for one point (terrain_manager.py:100-166): the in-place normalisation chain, then
// F.grid_sample(bilinear, border, align_corners=True) on the [H,W] field.  One rounding per op, taps summed left to right
// (nw, ne, sw, se); the oracle restates exactly this sequence.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float terrain_height(const GfTerrainView& tv, float x, float y) {
    if (!tv.height_field) return tv.origin_z;
    float nx = x - tv.x_min;
    nx = nx / tv.x_span;
    nx = nx * 2.0f;
    nx = nx - 1.0f;
    float ny = y - tv.y_min;
    ny = ny / tv.y_span;
    ny = ny * 2.0f;
    ny = ny - 1.0f;
    const int W = tv.cols, H = tv.rows;
    float ix = ((nx + 1.0f) / 2.0f) * (float)(W - 1);
    float iy = ((ny + 1.0f) / 2.0f) * (float)(H - 1);
    ix = clamp_max(clamp_min(ix, 0.0f), (float)(W - 1));
    iy = clamp_max(clamp_min(iy, 0.0f), (float)(H - 1));
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
    const float nw = (fx1 - ix) * (fy1 - iy);
    const float ne = (ix - fx0) * (fy1 - iy);
    const float sw = (fx1 - ix) * (iy - fy0);
    const float se = (ix - fx0) * (iy - fy0);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const bool x1_in = x0 + 1 < W, y1_in = y0 + 1 < H;
    const int x1 = x1_in ? x0 + 1 : x0, y1 = y1_in ? y0 + 1 : y0;
    const GF_GLOBAL float* f = G(tv.height_field);
    const float v_nw = f[y0 * W + x0];
    const float v_ne = x1_in ? f[y0 * W + x1] : 0.0f;
    const float v_sw = y1_in ? f[y1 * W + x0] : 0.0f;
    const float v_se = (x1_in && y1_in) ? f[y1 * W + x1] : 0.0f;
    return ((v_nw * nw + v_ne * ne) + v_sw * sw) + v_se * se;
}

// sin and cos with a fixed sequence of f32 operations (3-part Cody-Waite reduction by pi/2, Cephes minimax polynomials on
// [-pi/4, pi/4]), so the quaternion a reset writes is bit-identical on the GPU and in the oracle; within 2 ulp of libm.
__device__ __forceinline__ void sincos_det(float x, float* s, float* c) {
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

// genesis.utils.geom.xyz_to_quat restated (mdp/reset.py:63,194): extrinsic x-y-z Euler angles (radians) -> (w,x,y,z)
__device__ __forceinline__ float4 xyz_to_quat_det(float ax, float ay, float az) {
    float sx, cx, sy, cy, sz, cz;
    sincos_det(ax * 0.5f, &sx, &cx);
    sincos_det(ay * 0.5f, &sy, &cy);
    sincos_det(az * 0.5f, &sz, &cz);
    float4 q;
    q.x = (cx * cy) * cz + (sx * sy) * sz;
    q.y = (sx * cy) * cz - (cx * sy) * sz;
    q.z = (cx * sy) * cz + (sx * cy) * sz;
    q.w = (cx * cy) * sz - (sx * sy) * cz;
    return q;
}

// mdp.reset.randomize_terrain_position for one env (mdp/reset.py:199-226 -> terrain_manager.py:170-279); `A` exposes the
// spawn_* fields and the terrain view of GfResetArgs.  `u` holds the five unit draws (x, y, rot x, rot y, rot z).
template <class A>
__device__ __forceinline__ void spawn_pose(const A& a, const float (&u)[5], float (&pos)[3], float4* quat) {
    pos[0] = u[0] * a.spawn_x_span + a.spawn_x_min;
    pos[1] = u[1] * a.spawn_y_span + a.spawn_y_min;
    pos[2] = terrain_height(a.terrain, pos[0], pos[1]) + a.spawn_height_offset;
    const int m = a.spawn_rot_mask;
    const float rx = (m & 1) ? uniform_range(u[2], a.spawn_rot_lo[0], a.spawn_rot_hi[0]) : 0.0f;
    const float ry = (m & 2) ? uniform_range(u[3], a.spawn_rot_lo[1], a.spawn_rot_hi[1]) : 0.0f;
    const float rz = (m & 4) ? uniform_range(u[4], a.spawn_rot_lo[2], a.spawn_rot_hi[2]) : 0.0f;
    *quat = xyz_to_quat_det(rx, ry, rz);
}

// the five spawn draws of env `genv`: dense parity draws, or Philox blocks GF_SPAWN_BLOCK (x, y, rot z) and +1 (rot x, rot y)
__device__ __forceinline__ void spawn_draws(const float* draws, int64_t n, uint64_t seed, uint64_t stream, uint32_t genv, int rot_mask, float (&u)[5]) {
    if (draws) {
#pragma unroll
        for (int j = 0; j < 5; ++j) u[j] = draws[n * 5 + j];
        return;
    }
    const U4 r0 = philox4x32_10(genv, GF_SPAWN_BLOCK, (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
    u[0] = u24_to_unit(r0.x); u[1] = u24_to_unit(r0.y); u[4] = u24_to_unit(r0.z);
    u[2] = 0.0f; u[3] = 0.0f;
    if (rot_mask & 3) {
        const U4 r1 = philox4x32_10(genv, GF_SPAWN_BLOCK + 1u, (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed, (uint32_t)(seed >> 32));
        u[2] = u24_to_unit(r1.x); u[3] = u24_to_unit(r1.y);
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-level reductions (64 lanes) used only for the logging statistics.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, GF_WAVE);
    return v;  // valid in lane 0
}

__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }

// this workgroup's statistics shard (see GF_STATS_SHARDS in gf_step.h)
__device__ __forceinline__ GfStepStats* stats_shard(GfStepStats* s) { return s + (blockIdx.x % GF_STATS_SHARDS); }

// Fold the GF_STATS_SHARDS shards of one statistics slot into the GF_STATS_VECTOR_LEN-entry f64 vector (layout of
// gf_stats_pack): shards add, flag entries fold with max.  lane = shard, each entry is reduced across a wave with
// shuffles (a lane-per-entry loop over the shards would be 64 dependent memory round trips).
__device__ __forceinline__ double stats_entry(const GfStepStats& b, int v) {
    if (v < GF_MAX_TERM_TERMS) return (double)b.term_fired[v];
    if (v == GF_MAX_TERM_TERMS) return (double)b.reset_count;
    if (v == GF_MAX_TERM_TERMS + 1) return (double)(b.action_flags & 1);
    if (v == GF_MAX_TERM_TERMS + 2) return (double)((b.action_flags >> 1) & 1);
    if (v == GF_MAX_TERM_TERMS + 3) return (double)(b.contact_flags & 1);
    if (v == GF_MAX_TERM_TERMS + 4) return (double)b.resample_count;
    if (v < GF_MAX_TERM_TERMS + 5 + GF_MAX_TERMS) return b.reward_episode_sum[v - (GF_MAX_TERM_TERMS + 5)];
    if (v < GF_STATS_VECTOR_LEN) return (double)b.gait_count[v - (GF_MAX_TERM_TERMS + 5 + GF_MAX_TERMS)];
    return 0.0;
}

// One entry, one wave: lane = shard, shuffle tree.  (The tree order is part of the result for the f64 reward sums; every
// caller folds with this function so a statistic reads the same whichever kernel folded it.)
__device__ __forceinline__ void fold_stats_entry(const GfStepStats* src, double* dst, double* last_reset, int v) {
    static_assert(GF_STATS_SHARDS == GF_WAVE, "one lane per shard");
    const int lane = threadIdx.x & (GF_WAVE - 1);
    const GfStepStats& b = src[lane];
    double r = stats_entry(b, v);
    const double resets = last_reset ? wave_sum((double)b.reset_count) : 0.0;  // valid in lane 0
    const bool is_flag = v > GF_MAX_TERM_TERMS && v < GF_MAX_TERM_TERMS + 4;
    if (is_flag) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_down(r, off, GF_WAVE);
            r = o > r ? o : r;
        }
    } else {
        r = wave_sum(r);
    }
    if (lane == 0 && v < GF_STATS_VECTOR_LEN) {
        dst[v] = r;
        if (last_reset && resets > 0.0) last_reset[v] = r;
    }
}

__device__ __forceinline__ void fold_stats_block256(const GfStepStats* src, double* dst, double* last_reset) {
    for (int v = (int)(threadIdx.x / GF_WAVE); v < GF_STATS_VECTOR_LEN; v += 4) fold_stats_entry(src, dst, last_reset, v);
}

// four [3]-rows of a per-link array, links l0 … l0+3 of L: every load issued before the first use (a link at a time was a round trip
// per link — the link count is a run-time value, so the loop over it is not unrolled); rows past the last link re-read it
__device__ __forceinline__ void link_rows4(float (&f)[4][3], const GF_GLOBAL float* r, int l0, int L) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int l = l0 + j < L ? l0 + j : L - 1;
        f[j][0] = r[3 * l]; f[j][1] = r[3 * l + 1]; f[j][2] = r[3 * l + 2];
    }
}

// contact predicates shared by termination / reward terms
__device__ __forceinline__ int contact_count_over(const GfContactView& v, int64_t n, float thr) {
    int cnt = 0;
    const int L = v.num_links;
    const GF_GLOBAL float* r = G(v.contacts) + n * L * 3;
    for (int l0 = 0; l0 < L; l0 += 4) {
        float f[4][3];
        link_rows4(f, r, l0, L);
#pragma unroll
        for (int j = 0; j < 4; ++j) cnt += (l0 + j < L && norm3(f[j][0], f[j][1], f[j][2]) > thr) ? 1 : 0;
    }
    return cnt;
}

}  // namespace gf
