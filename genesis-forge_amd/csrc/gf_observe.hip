// gf_observe.hip — Phase O: ObservationManager.get_observations as one launch.
//
// Replaces managers/observation_manager.py:218-256 (per item: fn(), in-place *= scale,
// += uniform_(-1,1)*noise, then torch.cat; history list pop/insert + another cat: 29 aten ops /
// 23 launches for the 7-item Go2 config) and the mdp/observations.py getters.
//
// One 256-thread workgroup owns a tile of 64 consecutive envs.  The tile's [64, O] frame is assembled in LDS
// (row stride O+1 words, so the row-per-lane writes of the entity items and the flat cooperative
// copies of the [N,w] items are both conflict-free), then streamed out with full-width coalesced
// stores — the [N,O] row-major result is written exactly once, never re-read.
//   * [N,w] sources (commands, dof_pos, dof_vel, dof_force, targets, raw actions, external columns)
//     are contiguous over the tile: lanes copy them flat, 16 B per lane when w % 4 == 0.
//   * body-frame vectors (ang vel, lin vel, projected gravity) are computed once per env by the
//     env's own lane from quat/vel/ang loaded up front.
//   * scale and noise are applied on the way into LDS; noise is Philox(seed, stream, env, column)
//     or a caller-supplied dense U[0,1) array (parity mode).
//   * history (H > 1): the reference keeps a Python list of H tensors and re-concatenates them each
//     step, newest first.  Here the previous output is the history: frame slots 1..H-1 of the new
//     output are the previous output's slots 0..H-2 (flat coalesced copy), slot 0 is the new frame.
// Algorithmic traffic (Go2 command config, O=48, H=1): R 196 + W 192 = 388 B/env.
#include "gf_launch.h"

namespace gf {

enum : uint32_t { ON_QUAT = 1, ON_LIN = 2, ON_ANG = 4 };

// Four waves share a 64-env tile: wave 0 computes the per-env (body-frame) items, all 256 lanes do the flat copies into the
// LDS tile, the write-out and the history shift.  (One wave per tile left a 310-wide frame with 300 dependent
// load → store round trips per lane.)
constexpr int kObsBlock = 256;

// floor(i / d) by multiply-shift with m = ceil(2^40 / d): exact for i < 2^40 / d (here i < 64·d and d < 2^17) — the flat copies
// turn an element index into (row, column) once per element, and d (an item or frame width) is a run-time value
struct FastDiv {
    uint64_t m;
    uint32_t d;
    __device__ __forceinline__ explicit FastDiv(int div) : m(div > 1 ? ((1ull << 40) + (uint64_t)div - 1ull) / (uint64_t)div : 0ull), d((uint32_t)div) {}
    __device__ __forceinline__ int div(int i) const { return d > 1 ? (int)(((uint64_t)(uint32_t)i * m) >> 40) : i; }
};

struct TileCtx {
    float* tile;      // LDS, [64][S]
    int S;            // row stride in words (O+1)
    int64_t n0;       // first env of the tile
    int rows;         // live rows in the tile (<= 64)
    int tid;
    const GfObservationArgs* a;
};

__device__ __forceinline__ float finish(const GfObservationArgs& a, const GfObsItem& it, float v, int64_t n, int col) {
    if (it.scale != 1.0f) v = v * it.scale;  // observation_manager.py:242-244
    if (it.noise != 0.0f) {                  // observation_manager.py:247-250
        const float u = draw_u(a.noise_draws, n * a.obs_width + col, a.seed, a.stream, (uint32_t)n + a.env_offset, (uint32_t)col);
        v = v + uniform_range(u, -1.0f, 1.0f) * it.noise;
    }
    return v;
}

// Flat cooperative copy of a [rows, w] source block (row stride = src_stride words) into tile columns [col0, col0+w).
__device__ __forceinline__ void copy_rows(const int V, const TileCtx& c, const GfObsItem& it, const float* __restrict__ src, int src_stride, int w, int col0) {
    const GfObservationArgs& a = *c.a;
    const float* base = src + c.n0 * src_stride;
    const int total = c.rows * w;
    const FastDiv dw(w);
    int done = 0;
    if (V == 4 && src_stride == w && (reinterpret_cast<uintptr_t>(base) & 15u) == 0) {
        const int total4 = total >> 2;
        for (int i = c.tid; i < total4; i += kObsBlock) {
            const float4 x = reinterpret_cast<const float4*>(base)[i];
            const float xs[4] = {x.x, x.y, x.z, x.w};
            const int e = i * 4;
            int row = dw.div(e), cc = e - row * w;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c.tile[row * c.S + col0 + cc] = finish(a, it, xs[j], c.n0 + row, col0 + cc);
                if (++cc == w) { cc = 0; ++row; }
            }
        }
        done = total4 << 2;  // ragged tail (partial tile, w % 4 != 0) falls through to the scalar loop
    }
    for (int i = done + c.tid; i < total; i += kObsBlock) {
        const int row = dw.div(i), cc = i - row * w;
        c.tile[row * c.S + col0 + cc] = finish(a, it, base[row * src_stride + cc], c.n0 + row, col0 + cc);
    }
}

// History shift: frame slots 1..H-1 of the new rows are slots 0..H-2 of the previous output.  `hw` units per row, rows
// OHu units apart, `T` = float4 / float2 / float.  Four independent loads are in flight per lane before the first store.
template <typename T>
__device__ __forceinline__ void shift_history(T* __restrict__ dst, const T* __restrict__ src, int rows, int hw, int64_t OHu, int tid) {
    const FastDiv dh(hw);
    const int total = rows * hw;
    int i = tid;
    for (; i + 3 * kObsBlock < total; i += 4 * kObsBlock) {
        T v[4];
        int64_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int e = i + k * kObsBlock, row = dh.div(e);
            o[k] = (int64_t)row * OHu + (e - row * hw);
            v[k] = src[o[k]];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[o[k]] = v[k];
    }
    for (; i < total; i += kObsBlock) {
        const int row = dh.div(i);
        const int64_t o = (int64_t)row * OHu + (i - row * hw);
        dst[o] = src[o];
    }
}

// V = floats per memory operation of the write-out and the history shift: 4 when O % 4 == 0, 2 when O is even, else 1.
// A run-time value here (wave-uniform branches); the stand-alone kernels pass a constant, which folds the branches away.
__device__ __forceinline__ void observe_body(const int V, const GfObservationArgs& a, const uint32_t needs, float* tile) {
    const int tid = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * kEnvBlock;
    const int64_t N = a.num_envs;
    const int rows = (int)((N - n0) < kEnvBlock ? (N - n0) : kEnvBlock);
    const int O = a.obs_width, H = a.history_len, S = O + 1, D = a.num_dofs;
    const bool owner = tid < kEnvBlock;      // wave 0: lane = env
    const bool live = owner && tid < rows;
    const int64_t n = live ? n0 + tid : n0;

    TileCtx c{tile, S, n0, rows, tid, &a};

    // per-env entity state, requested before anything else (wave 0 only; the other waves read the zero pad)
    const uint32_t e = (uint32_t)n;
    // quirk: envs reset in this tick are rotated by their pre-reset quaternion (entity_manager.py:189-195)
    const bool use_stale = a.stale_quat != nullptr;
    const int s1 = *gsel(owner && use_stale && a.stale_mask, a.stale_mask, e);
    const int s2 = *gsel(owner && use_stale && a.stale_mask2, a.stale_mask2, e);
    const float* qsrc = (use_stale && (s1 | s2)) ? a.stale_quat : a.entity.quat;
    const float4 q = ldg4(gsel(owner && (needs & ON_QUAT) != 0, qsrc, 4u * e));
    const GF_GLOBAL float* lp = gsel(owner && (needs & ON_LIN) != 0, a.entity.lin_vel, 3u * e);
    const GF_GLOBAL float* ap = gsel(owner && (needs & ON_ANG) != 0, a.entity.ang_vel, 3u * e);
    const V3 lin{lp[0], lp[1], lp[2]}, ang{ap[0], ap[1], ap[2]};

    int col = 0;
    for (int i = 0; i < a.num_items; ++i) {
        const GfObsItem& it = a.items[i];
        const int w = it.width;
        switch (it.op) {
            case GF_O_COMMAND: copy_rows(V, c, it, a.command[it.i0].command, cmd_stride(a.command[it.i0]), w, col); break;
            case GF_O_DOF_POS: copy_rows(V, c, it, a.dof_pos, D, w, col); break;
            case GF_O_DOF_VEL: copy_rows(V, c, it, a.dof_vel, D, w, col); break;
            case GF_O_DOF_FORCE: copy_rows(V, c, it, a.dof_force, D, w, col); break;
            case GF_O_ACTIONS: copy_rows(V, c, it, a.targets, D, w, col); break;
            case GF_O_RAW_ACTIONS: copy_rows(V, c, it, a.env_actions, D, w, col); break;
            case GF_O_EXTERNAL: copy_rows(V, c, it, a.ext[it.i0], w, w, col); break;
            case GF_O_BASE_POS: copy_rows(1, c, it, a.entity.pos, 3, 3, col); break;
            case GF_O_BASE_QUAT: copy_rows(V, c, it, a.entity.quat, 4, 4, col); break;
            case GF_O_ANG_VEL_BODY:
            case GF_O_LIN_VEL_BODY:
            case GF_O_PROJ_GRAVITY: {
                if (live) {
                    const V3 v = it.op == GF_O_ANG_VEL_BODY ? rot_inv(q, ang) : (it.op == GF_O_LIN_VEL_BODY ? rot_inv(q, lin) : rot_inv(q, V3{0.f, 0.f, -1.f}));
                    float* r = tile + tid * S + col;
                    r[0] = finish(a, it, v.x, n, col);
                    r[1] = finish(a, it, v.y, n, col + 1);
                    r[2] = finish(a, it, v.z, n, col + 2);
                }
            } break;
            case GF_O_CONTACT_FORCE_NORM: {
                const GfContactView& cv = a.contact[it.i0];
                if (live) {
                    const float* r = cv.contacts + n * cv.num_links * 3;
                    for (int l = 0; l < w; ++l) tile[tid * S + col + l] = finish(a, it, norm3(r[3 * l], r[3 * l + 1], r[3 * l + 2]), n, col + l);
                }
            } break;
            default: break;
        }
        col += w;
    }

    const int64_t OH = (int64_t)O * H;
    float* out = a.obs + n0 * OH;
    // the history shift does not depend on the tile: issue it before the barrier so its loads overlap the assembly
    if (H > 1) {
        const float* prev = a.prev_obs + n0 * OH;
        const int hw = O * (H - 1);
        if (V == 4) shift_history(reinterpret_cast<float4*>(out + O), reinterpret_cast<const float4*>(prev), rows, hw >> 2, OH >> 2, tid);
        else if (V == 2) shift_history(reinterpret_cast<float2*>(out + O), reinterpret_cast<const float2*>(prev), rows, hw >> 1, OH >> 1, tid);
        else shift_history(out + O, prev, rows, hw, OH, tid);
    }
    __syncthreads();

    if (V == 4) {
        const int o4 = O >> 2;
        const FastDiv d4(o4);
        for (int i = tid; i < rows * o4; i += kObsBlock) {
            const int row = d4.div(i), c4 = i - row * o4;
            const float* r = tile + row * S + c4 * 4;
            reinterpret_cast<float4*>(out + row * OH)[c4] = make_float4(r[0], r[1], r[2], r[3]);
        }
    } else if (V == 2) {
        const int o2 = O >> 1;
        const FastDiv d2(o2);
        for (int i = tid; i < rows * o2; i += kObsBlock) {
            const int row = d2.div(i), c2 = i - row * o2;
            const float* r = tile + row * S + c2 * 2;
            reinterpret_cast<float2*>(out + row * OH)[c2] = make_float2(r[0], r[1]);
        }
    } else {
        const FastDiv d1(O);
        for (int i = tid; i < rows * O; i += kObsBlock) {
            const int row = d1.div(i), cc = i - row * O;
            out[row * OH + cc] = tile[row * S + cc];
        }
    }
}

#ifndef GF_BODIES_ONLY
template <int V>
__global__ __launch_bounds__(kObsBlock) void observe_kernel(const GfObservationArgs a, const uint32_t needs) {
    prefetch_args<GfObservationArgs>();
    extern __shared__ __attribute__((aligned(16))) float tile[];
    observe_body(V, a, needs, tile);
}
#endif

}  // namespace gf

#ifndef GF_BODIES_ONLY
namespace gf {
// validation, the entity inputs the items need, and the widest memory operation the frame width / alignment allow
int observe_prep(const GfObservationArgs* a, uint32_t* needs_out, int* vec_out) {
    if (!a || !a->obs) return GF_E_NULL;
    if (a->num_items <= 0 || a->num_items > GF_MAX_OBS_ITEMS || a->num_envs < 0) return GF_E_RANGE;
    if (a->history_len < 1) return GF_E_RANGE;
    if (a->history_len > 1 && (!a->prev_obs || a->prev_obs == a->obs)) return GF_E_NULL;
    const int O = a->obs_width, D = a->num_dofs;
    if (O <= 0 || O >= GF_MAX_OBS_WIDTH) return GF_E_RANGE;
    int wsum = 0;
    uint32_t needs = 0;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    auto al8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
    const bool vec4 = (O % 4 == 0) && al16(a->obs) && (a->history_len == 1 || al16(a->prev_obs));
    const bool vec2 = (O % 2 == 0) && al8(a->obs) && (a->history_len == 1 || al8(a->prev_obs));
    for (int i = 0; i < a->num_items; ++i) {
        const GfObsItem& it = a->items[i];
        if (it.width <= 0) return GF_E_RANGE;
        const float* src = nullptr;
        int stride = 0;
        switch (it.op) {
            case GF_O_COMMAND:
                if (it.i0 < 0 || it.i0 >= GF_MAX_COMMAND_VIEWS || !a->command[it.i0].command) return GF_E_SLOT;
                if (it.width != a->command[it.i0].width) return GF_E_RANGE;
                src = a->command[it.i0].command; stride = a->command[it.i0].stride ? a->command[it.i0].stride : it.width;
                break;
            case GF_O_DOF_POS: src = a->dof_pos; stride = D; break;
            case GF_O_DOF_VEL: src = a->dof_vel; stride = D; break;
            case GF_O_DOF_FORCE: src = a->dof_force; stride = D; break;
            case GF_O_ACTIONS: src = a->targets; stride = D; break;
            case GF_O_RAW_ACTIONS: src = a->env_actions; stride = D; break;
            case GF_O_EXTERNAL:
                if (it.i0 < 0 || it.i0 >= GF_MAX_EXT || !a->ext[it.i0]) return GF_E_SLOT;
                src = a->ext[it.i0]; stride = it.width;
                break;
            case GF_O_BASE_POS:
                if (!a->entity.pos) return GF_E_NULL;
                if (it.width != 3) return GF_E_RANGE;
                break;
            case GF_O_BASE_QUAT:
                if (it.width != 4) return GF_E_RANGE;
                src = a->entity.quat; stride = 4;
                break;
            case GF_O_ANG_VEL_BODY: needs |= gf::ON_QUAT | gf::ON_ANG; if (it.width != 3) return GF_E_RANGE; break;
            case GF_O_LIN_VEL_BODY: needs |= gf::ON_QUAT | gf::ON_LIN; if (it.width != 3) return GF_E_RANGE; break;
            case GF_O_PROJ_GRAVITY: needs |= gf::ON_QUAT; if (it.width != 3) return GF_E_RANGE; break;
            case GF_O_CONTACT_FORCE_NORM:
                if (it.i0 < 0 || it.i0 >= GF_MAX_CONTACT_VIEWS || !a->contact[it.i0].contacts) return GF_E_SLOT;
                if (it.width != a->contact[it.i0].num_links) return GF_E_RANGE;
                break;
            default: return GF_E_OPCODE;
        }
        if (stride) {
            if (!src) return GF_E_NULL;
            if (it.width > stride) return GF_E_RANGE;
        }
        wsum += it.width;
    }
    if (wsum != O) return GF_E_RANGE;
    if ((needs & gf::ON_QUAT) && (!a->entity.quat || !al16(a->entity.quat))) return a->entity.quat ? GF_E_UNSUPPORTED : GF_E_NULL;
    if ((needs & gf::ON_LIN) && !a->entity.lin_vel) return GF_E_NULL;
    if ((needs & gf::ON_ANG) && !a->entity.ang_vel) return GF_E_NULL;
    *needs_out = needs;
    *vec_out = vec4 ? 4 : (vec2 ? 2 : 1);
    return GF_OK;
}
}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_observe(const GfObservationArgs* a, void* stream) {
    uint32_t needs = 0;
    int vec = 1;
    const int rc = gf::observe_prep(a, &needs, &vec);
    if (rc) return rc;
    if (a->num_envs == 0) return GF_OK;
    const bool vec4 = vec == 4, vec2 = vec == 2;
    const int O = a->obs_width;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)(O + 1) * gf::kEnvBlock * sizeof(float);
    gf::PhaseScope scope(GF_PHASE_OBSERVE, s);
    scope.begin_bracket();
    const unsigned grid = gf::env_grid(a->num_envs);  // one 256-thread workgroup per 64-env tile
    if (vec4) gf::klaunch(gf::observe_kernel<4>, dim3(grid), dim3(gf::kObsBlock), lds, s, *a, needs);
    else if (vec2) gf::klaunch(gf::observe_kernel<2>, dim3(grid), dim3(gf::kObsBlock), lds, s, *a, needs);
    else gf::klaunch(gf::observe_kernel<1>, dim3(grid), dim3(gf::kObsBlock), lds, s, *a, needs);
    return gf::launch_status();
}
#endif
