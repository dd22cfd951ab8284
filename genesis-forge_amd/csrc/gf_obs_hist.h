// gf_obs_hist.h — the observation history shift, shared by the stand-alone observation kernel / chain B (gf_observe.hip) and
// the fused post-physics kernel (gf_post_ws.h): both run 256-thread workgroups over 64-env tiles.
#pragma once

#include "gf_device.h"

namespace gf {

constexpr int kObsBlock = 256;  // threads of a workgroup that owns one 64-env observation tile
constexpr int kObsShift = 8;    // history units per lane in flight together

// floor(i / d) by multiply-shift with m = ceil(2^40 / d): exact for i < 2^40 / d (here i < 64·d and d < 2^17) — an element index
// becomes (row, column) once per element, and d (a frame or history width) is a run-time value
struct FastDiv {
    uint64_t m;
    uint32_t d;
    __device__ __forceinline__ explicit FastDiv(int div) : m(div > 1 ? ((1ull << 40) + (uint64_t)div - 1ull) / (uint64_t)div : 0ull), d((uint32_t)div) {}
    __device__ __forceinline__ int div(int i) const { return d > 1 ? (int)(((uint64_t)(uint32_t)i * m) >> 40) : i; }
};

typedef float f32x2a __attribute__((ext_vector_type(2)));               // 8-byte aligned
typedef float f32x4a __attribute__((ext_vector_type(4)));               // 16-byte aligned
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // dword aligned: one global_load_dwordx4 all the same

// History shift.  The [rows, O·H] block of a tile is one contiguous run of floats in `out` and in `prev`, and
// out[k] = prev[k - O] wherever column (k mod O·H) >= O.  Cut the run into 16-byte units aligned on `out`: a unit that lies
// entirely in history columns is one (dword-aligned) 16-byte load and one aligned 16-byte store, whatever O is — rows of an
// odd-width frame are not 16-byte aligned, the run is.  Units that touch a new-frame column wait for the LDS tile
// (write_mixed_units).  A batch = kObsShift units per lane, loads first, stores later: the caller puts other work between
// the two so the lane never sits on an empty queue.
// `NT`: the non-temporal hint on the loads of the old history and the stores of the new tensor, for outputs no cache keeps from one
// step to the next (kObsStreamBytes, gf_post_args.h)
template <bool NT, class T>
__device__ __forceinline__ T obs_stream_load(const GF_GLOBAL T* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, class T>
__device__ __forceinline__ void obs_stream_store(GF_GLOBAL T* p, const T v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
struct HistBatch {
    f32x4u v[kObsShift];
    uint32_t pure;   // bit k: unit k of the batch is a pure history unit of this lane
};
// `stride` = lanes that share the run (the whole 256-thread workgroup, or the 192 lanes of the three waves that shift history while
// the fourth folds rewards in the fused kernel)
template <bool NT = false>
__device__ __forceinline__ void hist_load(HistBatch& b, const GF_GLOBAL float* __restrict__ prev, int first, int units, int O, int OH, const FastDiv& dr,
                                          const int stride = kObsBlock) {
    b.pure = 0u;
    // UNCONDITIONAL loads (a unit that is not this lane's to move re-reads the run's first unit): eight loads back to back, no
    // basic blocks in between.  With `if (pure) load` the compiler put every load and every store in a block of its own and — loads
    // and stores share vmcnt on gfx9 and complete out of order with respect to each other — an `s_waitcnt vmcnt(0)` in front of
    // EVERY store: eight serialised round trips per batch (ISA of r02's first gait kernel; 45 us for 164 MB at 65 536 envs).
#pragma unroll
    for (int k = 0; k < kObsShift; ++k) {
        const int u = first + k * stride, uu = u < units ? u : 0;
        const int e = uu << 2, row = dr.div(e), c = e - row * OH;
        const bool pure = u < units && c >= O && c + 3 < OH;
        b.pure |= pure ? 1u << k : 0u;
        b.v[k] = obs_stream_load<NT>(reinterpret_cast<const GF_GLOBAL f32x4u*>(prev + (pure ? e - O : 0)));
    }
}
template <bool NT = false>
__device__ __forceinline__ void hist_store(const HistBatch& b, GF_GLOBAL float* out, int first, const int stride = kObsBlock, const bool wait = true) {
    if (wait) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), once: every load of the batch has landed, the stores below need no further waits
#pragma unroll
    for (int k = 0; k < kObsShift; ++k)
        if ((b.pure >> k) & 1u) obs_stream_store<NT>(reinterpret_cast<GF_GLOBAL f32x4a*>(out + ((first + k * stride) << 2)), f32x4a{b.v[k].x, b.v[k].y, b.v[k].z, b.v[k].w});
}

// … and the units the history batches left: every unit with at least one new-frame column (frame from the LDS tile, the
// history elements it shares a unit with from `prev`), plus the run's last rows·O·H mod 4 floats.
// `out2` (optional, wave-uniform): a second destination with the same layout — the rollout-storage row of the RL library (§8f-5)
template <bool NT = false>
__device__ __forceinline__ void write_mixed_units(GF_GLOBAL float* __restrict__ out, const GF_GLOBAL float* __restrict__ prev, const float* tile, int S, int rows, int O,
                                                  int OH, int tid, GF_GLOBAL float* __restrict__ out2 = nullptr) {
    const int total = rows * OH, units = total >> 2;
    const int upr = ((O + 3) >> 2) + 1;  // units that can touch one row's frame columns
    const FastDiv du(upr);
    auto element = [&](int k, int r) -> float {
        const int rr = k >= r * OH ? r : r - 1, c = k - rr * OH;   // the unit's leading floats can be the previous row's history
        return c < O ? tile[rr * S + c] : prev[k - O];
    };
    constexpr int kU = 5;  // units per lane whose boundary loads are in flight together
    for (int i0 = tid; i0 < rows * upr; i0 += kU * kObsBlock) {
        f32x4a v[kU];
        int at[kU];
#pragma unroll
        for (int b = 0; b < kU; ++b) {
            const int i = i0 + b * kObsBlock;
            const int ii = i < rows * upr ? i : i0, r = du.div(ii), j = ii - r * upr;
            const int u = ((r * OH) >> 2) + j;
            // two frames are O·(H-1) >= 4 floats apart: a unit touches one frame at most, so each is written once
            const bool on = i < rows * upr && u <= ((r * OH + O - 1) >> 2) && u < units;
            at[b] = on ? u << 2 : -1;
            v[b] = f32x4a{0.f, 0.f, 0.f, 0.f};
            if (on) {
                const int c0 = (u << 2) - r * OH;   // the unit's first column in row r (negative: it starts in row r-1's history)
                if (c0 >= 0 && c0 + 3 < O) {        // entirely inside the new frame (all but <= 2 units of a row): the LDS tile only
                    const float* tp = tile + r * S + c0;
                    v[b] = f32x4a{tp[0], tp[1], tp[2], tp[3]};
                } else {                            // straddles a frame edge: the history elements come from `prev`
                    v[b] = f32x4a{element(u << 2, r), element((u << 2) + 1, r), element((u << 2) + 2, r), element((u << 2) + 3, r)};
                }
            }
        }
#pragma unroll
        for (int b = 0; b < kU; ++b)
            if (at[b] >= 0) obs_stream_store<NT>(reinterpret_cast<GF_GLOBAL f32x4a*>(out + at[b]), v[b]);
        if (out2) {
#pragma unroll
            for (int b = 0; b < kU; ++b)
                if (at[b] >= 0) obs_stream_store<NT>(reinterpret_cast<GF_GLOBAL f32x4a*>(out2 + at[b]), v[b]);
        }
    }
    const int tail = total & 3;
    if (tid < tail) {
        const int k = (units << 2) + tid, r = rows - 1, c = k - r * OH;
        const float v = c < O ? tile[r * S + c] : prev[k - O];
        out[k] = v;
        if (out2) out2[k] = v;
    }
}

}  // namespace gf
