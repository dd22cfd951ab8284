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

// NaN-propagating clamps with torch semantics (Appendix B of SURVEY.md)
__device__ __forceinline__ float clamp_max(float x, float hi) { return x > hi ? hi : x; }
__device__ __forceinline__ float clamp_min(float x, float lo) { return x < lo ? lo : x; }

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
    if (v < GF_STATS_VECTOR_LEN) return b.reward_episode_sum[v - (GF_MAX_TERM_TERMS + 5)];
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

// contact predicates shared by termination / reward terms
__device__ __forceinline__ int contact_count_over(const GfContactView& v, int64_t n, float thr) {
    int cnt = 0;
    const GF_GLOBAL float* r = G(v.contacts) + n * v.num_links * 3;
    for (int l = 0; l < v.num_links; ++l) cnt += norm3(r[3 * l], r[3 * l + 1], r[3 * l + 2]) > thr ? 1 : 0;
    return cnt;
}

}  // namespace gf
