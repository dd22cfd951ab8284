// gf_observe.hip — Phase O: ObservationManager.get_observations as one launch.
//
// Replaces managers/observation_manager.py:218-256 (per item: fn(), in-place *= scale,
// += uniform_(-1,1)*noise, then torch.cat; history list pop/insert + another cat: 29 aten ops /
// 23 launches for the 7-item Go2 config) and the mdp/observations.py getters.
//
// One 256-thread workgroup owns a tile of 64 consecutive envs.  The tile's [64, O] frame is assembled in LDS (row stride
// O+1 words: conflict-free for row-per-lane and column-per-lane writes alike), then streamed out with full-width coalesced
// stores — the [N,O] row-major result is written exactly once, never re-read.
//   * the item table is resolved once per workgroup, lane-parallel (lane i reads item i from the kernel-argument segment,
//     lane c finds the source of column c); a walk of the table with scalar loads costs one load latency per item;
//   * [N,w] sources (commands, dof_pos, dof_vel, dof_force, targets, raw actions, external columns, base pos / quat,
//     contact-force norms) are gathered column-per-lane: every load a lane needs is in flight before the first LDS write;
//   * body-frame vectors (ang vel, lin vel, projected gravity) are computed once per env by the env's own lane (wave 0) from
//     quat / vel / ang requested before anything else;
//   * scale and noise are applied on the way into LDS; noise is Philox(seed, stream, env, column) or a caller-supplied dense
//     U[0,1) array (parity mode);
//   * history (H > 1): the reference keeps a Python list of H tensors and re-concatenates them each step, newest first.
//     Here the previous output is the history: frame slots 1..H-1 of the new output are the previous output's slots 0..H-2,
//     moved as 16-byte units of the tile's contiguous run (HistBatch) in batches whose loads are issued before the table and
//     the gather are waited for, so that wait is spent with history traffic in flight.
// Algorithmic traffic (Go2 command config, O=48, H=1): R 196 + W 192 = 388 B/env.
#include "gf_launch.h"
#include "gf_obs_hist.h"

namespace gf {

enum : uint32_t { ON_QUAT = 1, ON_LIN = 2, ON_ANG = 4 };

// Four waves share a 64-env tile: wave 0 computes the per-env (body-frame) items, all 256 lanes gather the [N,w] items into
// the LDS tile, write it out and shift the history.
#define GF_OBS_INLINE __attribute__((always_inline))
constexpr int kObsGather = 8;   // gather units (one or four floats each) per lane whose loads are in flight together

// Where a frame's columns come from.  The item table is resolved ONCE per workgroup, lane-parallel: lane i reads item i
// straight from the kernel-argument segment (a vector load: no serial walk of scalar loads), lane c then finds the item that
// owns column c.  Both tables live in LDS because lanes of one wave look up different columns.
enum : int32_t { OS_ROWS = 0, OS_OWNER = 1, OS_NORM3 = 2 };
struct ObsItemRec {
    const float* src;   // OS_ROWS: element (n, j) = src[n·stride + j];  OS_NORM3: |src[n·stride + 3j .. 3j+2]|
    int32_t stride, width, kind, op;
    float scale, noise;
    int32_t units, vec;   // gather units of this item: width / 4 units of FOUR columns (vec) or `width` units of one; 0 for per-env items
};
// A gather unit: what ONE lane loads per row.  An item whose rows are 16-byte aligned runs of a multiple of four floats
// (dof_pos / dof_vel / targets / actions rows of 12 or 28 DOF, external columns of such widths) is gathered four columns at a
// time — one dwordx4 per (row, unit) instead of four dword loads: the Go2 frame's 39 gathered columns are 12 units.  Round 2
// measured this kernel at 0.29 of the HBM peak at 1 M envs against 0.67 for the fused kernel, which loads the same rows as float4s.
struct ObsUnit {
    const float* p;     // source of row 0's first element of this unit
    int32_t stride, kind, col0, w;   // w = 4 (one dwordx4) or 1
    float scale, noise;
};

// The stand-alone kernel gets the unit table resolved by the HOST (observe_prep) inside its kernel arguments: a lane reads its unit
// straight from the kernel-argument segment and wave 0 walks the per-env items with scalar loads — no LDS tables, none of the two
// barriers every workgroup otherwise spends before its first gather load is in flight.  Frames with more units than fit (or a
// phase chain, whose descriptors sit elsewhere in the argument segment) build the tables in LDS as before.
constexpr int kPlanUnits = 48, kPlanOwners = 8;
struct ObsOwner { int32_t op, col; float scale, noise; };
struct ObsPlan {
    int32_t num_units, num_owners;   // num_units == 0: no plan (build the tables in the kernel)
    ObsOwner owner[kPlanOwners];
    ObsUnit unit[kPlanUnits];
};
struct ObsKernelArgs {
    GfObservationArgs a;
    uint32_t needs, _pad;
    ObsPlan plan;
};
static_assert(sizeof(ObsKernelArgs) <= 4096, "kernarg segment");

__device__ __forceinline__ float finish(const GfObservationArgs& a, const float scale, const float noise, float v, int64_t n, int col) {
    if (scale != 1.0f) v = v * scale;  // observation_manager.py:242-244
    if (noise != 0.0f) {               // observation_manager.py:247-250
        const float u = draw_u(a.noise_draws, n * a.obs_width + col, a.seed, a.stream, (uint32_t)n + a.env_offset, (uint32_t)col);
        v = v + uniform_range(u, -1.0f, 1.0f) * noise;
    }
    return v;
}

// V = floats per memory operation of the frame write-out and the history shift: 4 when O % 4 == 0, 2 when O is even, else 1.
// A run-time value here (wave-uniform branches); the stand-alone kernels pass a constant, which folds the branches away.
__device__ __forceinline__ void observe_body(const int V, const GfObservationArgs& a, const uint32_t needs, float* tile, const uint32_t karg_off,
                                             const ObsPlan* plan = nullptr, const uint32_t plan_off = 0u) {
    __shared__ ObsItemRec s_item[GF_MAX_OBS_ITEMS];
    __shared__ ObsUnit s_unit[GF_MAX_OBS_WIDTH];
    const bool planned = plan != nullptr && plan->num_units > 0;   // wave-uniform (kernel argument)
    const int tid = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * kEnvBlock;
    const int64_t N = a.num_envs;
    const int rows = (int)((N - n0) < kEnvBlock ? (N - n0) : kEnvBlock);
    const int O = a.obs_width, H = a.history_len, S = O + 1, D = a.num_dofs;
    const int num_items = a.num_items;
    const bool owner = tid < kEnvBlock;      // wave 0: lane = env
    const bool live = owner && tid < rows;
    const int64_t n = live ? n0 + tid : n0;

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

    const int64_t OH = (int64_t)O * H;
    const bool ring = a.history_ring != 0;   // in-place ring: only the new frame is written, into its slot
    const int64_t OS = (ring && a.ring_slots) ? (int64_t)O * a.ring_slots : OH;   // env stride of `out` (a ring with more slots than frames)
    GF_GLOBAL float* out = G(a.obs) + n0 * OS + (ring ? (int64_t)(a.history_ring - 1) * O : 0);
    const GF_GLOBAL float* prev = (H > 1 && !ring) ? G(a.prev_obs) + n0 * OH : nullptr;
    // 16-byte units over the tile's contiguous run (see HistBatch); otherwise (a frame narrower than 4, an output that is not
    // 16-byte aligned) element by element
    const bool flat = H > 1 && !ring && O >= 4 && (reinterpret_cast<uintptr_t>(a.obs) & 15u) == 0;
    const bool hist = flat;
    const int units = (rows * (int)OH) >> 2;
    const FastDiv dr((int)OH);
    constexpr int kHistStep = kObsShift * kObsBlock;
    HistBatch hb;
    int hfirst = tid;
    // the history shift depends on nothing this kernel computes: its first batch of loads goes out before the item table is
    // even read, and every later wait (table, gather) is spent with history traffic in flight
    if (hist) hist_load(hb, prev, hfirst, units, O, (int)OH, dr);

    // item i → its source (lane i); `a` sits `karg_off` bytes into the kernel-argument segment
    if (!planned && tid < num_items) {
        const auto* kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
        const auto* ip = (const __attribute__((address_space(4))) int32_t*)(kp + karg_off + offsetof(GfObservationArgs, items) + (size_t)tid * sizeof(GfObsItem));
        const int op = ip[0], w = ip[1], i0 = ip[2];
        const float scale = __int_as_float(ip[4]), noise = __int_as_float(ip[5]);
        const float* src = nullptr;
        int stride = 0, kind = OS_ROWS;
        switch (op) {
            case GF_O_COMMAND:   // static slot loops: a lane-dependent index into the by-value descriptor would spill it to scratch
#pragma unroll
                for (int v = 0; v < GF_MAX_COMMAND_VIEWS; ++v)
                    if (i0 == v) { src = a.command[v].command; stride = cmd_stride(a.command[v]); }
                break;
            case GF_O_DOF_POS: src = a.dof_pos; stride = D; break;
            case GF_O_DOF_VEL: src = a.dof_vel; stride = D; break;
            case GF_O_DOF_FORCE: src = a.dof_force; stride = D; break;
            case GF_O_ACTIONS: src = a.targets; stride = D; break;
            case GF_O_RAW_ACTIONS: src = a.env_actions; stride = D; break;
            case GF_O_EXTERNAL:
#pragma unroll
                for (int v = 0; v < GF_MAX_EXT; ++v)
                    if (i0 == v) src = a.ext[v];
                stride = w;
                break;
            case GF_O_BASE_POS: src = a.entity.pos; stride = 3; break;
            case GF_O_BASE_QUAT: src = a.entity.quat; stride = 4; break;
            case GF_O_CONTACT_FORCE_NORM:
#pragma unroll
                for (int v = 0; v < GF_MAX_CONTACT_VIEWS; ++v)
                    if (i0 == v) { src = a.contact[v].contacts; stride = 3 * a.contact[v].num_links; }
                kind = OS_NORM3;
                break;
            default: kind = OS_OWNER; break;  // body-frame vectors: computed per env by wave 0
        }
        const int vec = (kind == OS_ROWS && (w & 3) == 0 && (stride & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) ? 1 : 0;
        s_item[tid] = ObsItemRec{src, stride, w, kind, op, scale, noise, kind == OS_OWNER ? 0 : (vec ? w >> 2 : w), vec};
    }
    if (!planned) __syncthreads();
    // unit u → its source (lane u): which item, which of its columns
    int U = planned ? plan->num_units : 0;
    if (!planned) {
        int cacc = 0, it = -1, ust = 0, cst = 0;
        for (int i = 0; i < num_items; ++i) {
            const int un = s_item[i].units;
            if (tid >= U && tid < U + un) { it = i; ust = U; cst = cacc; }
            U += un;
            cacc += s_item[i].width;
        }
        if (it >= 0) {
            const ObsItemRec r = s_item[it];
            const int j = tid - ust;
            s_unit[tid] = r.vec ? ObsUnit{r.src + 4 * j, r.stride, r.kind, cst + 4 * j, 4, r.scale, r.noise}
                                : ObsUnit{r.src + (r.kind == OS_NORM3 ? 3 * j : j), r.stride, r.kind, cst + j, 1, r.scale, r.noise};
        }
    }
    if (!planned) __syncthreads();

    // gather: every [N,w] element of the tile.  A lane keeps ONE unit (its source is looked up once) and walks rows: with
    // P = the power of two >= U (the number of units), lane t has unit t mod P and rows t/P, t/P + 256/P, … — consecutive lanes
    // read consecutive units of one row, the address advances by a constant, and the LDS writes (row stride O+1) are
    // conflict-free.  kObsGather loads per lane are in flight per pass; a pass = loads, then (later) the LDS writes, and the
    // history batches run between the two halves of the first pass.
    const int lgP = U > 1 ? 32 - __builtin_clz((unsigned)(U - 1)) : 0;
    const int gun = tid & ((1 << lgP) - 1), grow0 = tid >> lgP, grstep = kObsBlock >> lgP;
    ObsUnit gc;
    if (planned) {   // this lane's unit, straight from the kernel-argument segment (a vector load: the index is lane-dependent)
        const auto* kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
        typedef int32_t i32x4k __attribute__((ext_vector_type(4)));
        static_assert(sizeof(ObsUnit) == 32 && offsetof(ObsUnit, stride) == 8 && offsetof(ObsUnit, scale) == 24, "two 16-byte loads");
        const auto* up = (const __attribute__((address_space(4))) i32x4k*)(kp + plan_off + offsetof(ObsPlan, unit) + (size_t)(gun < U ? gun : 0) * sizeof(ObsUnit));
        const i32x4k w0 = up[0], w1 = up[1];
        gc.p = reinterpret_cast<const float*>(((uint64_t)(uint32_t)w0.y << 32) | (uint64_t)(uint32_t)w0.x);
        gc.stride = w0.z; gc.kind = w0.w; gc.col0 = w1.x; gc.w = w1.y;
        gc.scale = __int_as_float(w1.z); gc.noise = __int_as_float(w1.w);
    } else {
        gc = s_unit[gun < U ? gun : 0];
    }
    const bool gactive = gun < U;
    const bool gvec = gactive && gc.w == 4;
    const int64_t gstep = (int64_t)grstep * gc.stride;
    f32x4 gx[kObsGather];
    auto gather_load = [&](int row) GF_OBS_INLINE {
        const float* p = gc.p + (n0 + row) * gc.stride;
#pragma unroll
        for (int u = 0; u < kObsGather; ++u) {
            gx[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (gactive && row + u * grstep < rows) {
                if (gvec) gx[u] = *reinterpret_cast<const GF_GLOBAL f32x4*>(G(p + u * gstep));
                else gx[u].x = p[u * gstep];
            }
        }
    };
    auto gather_store = [&](int row) GF_OBS_INLINE {
#pragma unroll
        for (int u = 0; u < kObsGather; ++u) {
            const int r = row + u * grstep;
            if (!(gactive && r < rows)) continue;
            float* t = tile + r * S + gc.col0;
            if (gvec) {
                t[0] = finish(a, gc.scale, gc.noise, gx[u].x, n0 + r, gc.col0);
                t[1] = finish(a, gc.scale, gc.noise, gx[u].y, n0 + r, gc.col0 + 1);
                t[2] = finish(a, gc.scale, gc.noise, gx[u].z, n0 + r, gc.col0 + 2);
                t[3] = finish(a, gc.scale, gc.noise, gx[u].w, n0 + r, gc.col0 + 3);
            } else {
                float v = gx[u].x;
                if (gc.kind == OS_NORM3) {  // the other two components share the first one's cache line
                    const float* q3 = gc.p + (n0 + r) * gc.stride;
                    v = norm3(v, q3[1], q3[2]);
                }
                t[0] = finish(a, gc.scale, gc.noise, v, n0 + r, gc.col0);
            }
        }
    };
    gather_load(grow0);
    if (hist) {
        for (;;) {
            hist_store(hb, out, hfirst);
            hfirst += kHistStep;
            if (hfirst - tid >= units) break;  // wave-uniform: the batch's first unit index of lane 0
            hist_load(hb, prev, hfirst, units, O, (int)OH, dr);
        }
    } else if (H > 1 && !ring) {
        const int hw = O * (H - 1);
        const FastDiv dh(hw);
        for (int i = tid; i < rows * hw; i += kObsBlock) {
            const int row = dh.div(i);
            const int64_t o = (int64_t)row * OH + (i - row * hw);
            out[O + o] = prev[o];
        }
    }
    gather_store(grow0);
    for (int row = grow0 + kObsGather * grstep; row - grow0 < rows; row += kObsGather * grstep) {  // the rows one pass does not reach
        gather_load(row);
        gather_store(row);
    }
    // per-env items (their inputs were requested first and have long landed)
    if (live && needs && planned) {
#pragma unroll
        for (int j = 0; j < kPlanOwners; ++j) {
            if (j >= plan->num_owners) break;
            const ObsOwner it = plan->owner[j];
            const V3 v = it.op == GF_O_ANG_VEL_BODY ? rot_inv(q, ang) : (it.op == GF_O_LIN_VEL_BODY ? rot_inv(q, lin) : rot_inv(q, V3{0.f, 0.f, -1.f}));
            float* r = tile + tid * S + it.col;
            r[0] = finish(a, it.scale, it.noise, v.x, n, it.col);
            r[1] = finish(a, it.scale, it.noise, v.y, n, it.col + 1);
            r[2] = finish(a, it.scale, it.noise, v.z, n, it.col + 2);
        }
    } else if (live && needs) {
        int col = 0;
        for (int i = 0; i < num_items; ++i) {
            const ObsItemRec it = s_item[i];
            if (it.op == GF_O_ANG_VEL_BODY || it.op == GF_O_LIN_VEL_BODY || it.op == GF_O_PROJ_GRAVITY) {
                const V3 v = it.op == GF_O_ANG_VEL_BODY ? rot_inv(q, ang) : (it.op == GF_O_LIN_VEL_BODY ? rot_inv(q, lin) : rot_inv(q, V3{0.f, 0.f, -1.f}));
                float* r = tile + tid * S + col;
                r[0] = finish(a, it.scale, it.noise, v.x, n, col);
                r[1] = finish(a, it.scale, it.noise, v.y, n, col + 1);
                r[2] = finish(a, it.scale, it.noise, v.z, n, col + 2);
            }
            col += it.width;
        }
    }
    __syncthreads();

    if (flat) {
        write_mixed_units(out, prev, tile, S, rows, O, (int)OH, tid);
    } else if (V == 4) {
        const int o4 = O >> 2;
        const FastDiv d4(o4);
        for (int i = tid; i < rows * o4; i += kObsBlock) {
            const int row = d4.div(i), c4 = i - row * o4;
            const float* r = tile + row * S + c4 * 4;
            reinterpret_cast<GF_GLOBAL f32x4a*>(out + row * OS)[c4] = f32x4a{r[0], r[1], r[2], r[3]};
        }
    } else if (V == 2) {
        const int o2 = O >> 1;
        const FastDiv d2(o2);
        for (int i = tid; i < rows * o2; i += kObsBlock) {
            const int row = d2.div(i), c2 = i - row * o2;
            const float* r = tile + row * S + c2 * 2;
            reinterpret_cast<GF_GLOBAL f32x2a*>(out + row * OS)[c2] = f32x2a{r[0], r[1]};
        }
    } else {
        const FastDiv d1(O);
        for (int i = tid; i < rows * O; i += kObsBlock) {
            const int row = d1.div(i), cc = i - row * O;
            out[row * OS + cc] = tile[row * S + cc];
        }
    }
}

#ifndef GF_BODIES_ONLY
template <int V>
__global__ __launch_bounds__(kObsBlock) void observe_kernel(const ObsKernelArgs k) {
    prefetch_args<GfObservationArgs>();
    extern __shared__ __attribute__((aligned(16))) float tile[];
    observe_body(V, k.a, k.needs, tile, (uint32_t)offsetof(ObsKernelArgs, a), &k.plan, (uint32_t)offsetof(ObsKernelArgs, plan));
}
#endif

}  // namespace gf

#ifndef GF_BODIES_ONLY
namespace gf {
// validation, the entity inputs the items need, and the widest memory operation the frame width / alignment allow
int observe_prep(const GfObservationArgs* a, uint32_t* needs_out, int* vec_out) {
    if (!a || !a->obs) return GF_E_NULL;
    if (a->num_items <= 0 || a->num_items > GF_MAX_OBS_ITEMS || a->num_envs < 0) return GF_E_RANGE;
    const bool ring = a->history_ring != 0;
    if (a->ring_slots && (!ring || (int64_t)a->ring_slots < a->history_len)) return GF_E_RANGE;
    if (a->history_len < 1 || a->history_ring < 0 || (int64_t)a->history_ring > (a->ring_slots ? (int64_t)a->ring_slots : (int64_t)a->history_len)) return GF_E_RANGE;
    if (a->history_len > 1 && !ring && (!a->prev_obs || a->prev_obs == a->obs)) return GF_E_NULL;
    const int O = a->obs_width, D = a->num_dofs;
    if (O <= 0 || O >= GF_MAX_OBS_WIDTH) return GF_E_RANGE;
    int wsum = 0;
    uint32_t needs = 0;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    auto al8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; };
    const bool vec4 = (O % 4 == 0) && al16(a->obs) && (a->history_len == 1 || ring || al16(a->prev_obs));
    const bool vec2 = (O % 2 == 0) && al8(a->obs) && (a->history_len == 1 || ring || al8(a->prev_obs));
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

namespace gf {
// The unit / per-env item tables of a validated descriptor, as the kernel would build them (observe_body) — resolved on the host.
static void observe_plan(const GfObservationArgs* a, ObsPlan* pl) {
    pl->num_units = pl->num_owners = 0;
    const int D = a->num_dofs;
    int col = 0, units = 0, owners = 0;
    for (int i = 0; i < a->num_items; ++i) {
        const GfObsItem& it = a->items[i];
        const float* src = nullptr;
        int stride = 0, kind = OS_ROWS;
        switch (it.op) {
            case GF_O_COMMAND: src = a->command[it.i0].command; stride = a->command[it.i0].stride ? a->command[it.i0].stride : a->command[it.i0].width; break;
            case GF_O_DOF_POS: src = a->dof_pos; stride = D; break;
            case GF_O_DOF_VEL: src = a->dof_vel; stride = D; break;
            case GF_O_DOF_FORCE: src = a->dof_force; stride = D; break;
            case GF_O_ACTIONS: src = a->targets; stride = D; break;
            case GF_O_RAW_ACTIONS: src = a->env_actions; stride = D; break;
            case GF_O_EXTERNAL: src = a->ext[it.i0]; stride = it.width; break;
            case GF_O_BASE_POS: src = a->entity.pos; stride = 3; break;
            case GF_O_BASE_QUAT: src = a->entity.quat; stride = 4; break;
            case GF_O_CONTACT_FORCE_NORM: src = a->contact[it.i0].contacts; stride = 3 * a->contact[it.i0].num_links; kind = OS_NORM3; break;
            default: kind = OS_OWNER; break;
        }
        if (kind == OS_OWNER) {
            if (owners >= kPlanOwners) return;
            pl->owner[owners++] = ObsOwner{it.op, col, it.scale, it.noise};
        } else {
            const bool vec = kind == OS_ROWS && (it.width & 3) == 0 && (stride & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
            const int n = vec ? it.width >> 2 : it.width;
            if (units + n > kPlanUnits) return;
            for (int j = 0; j < n; ++j)
                pl->unit[units++] = vec ? ObsUnit{src + 4 * j, stride, kind, col + 4 * j, 4, it.scale, it.noise}
                                        : ObsUnit{src + (kind == OS_NORM3 ? 3 * j : j), stride, kind, col + j, 1, it.scale, it.noise};
        }
        col += it.width;
    }
    if (units == 0) return;   // (a frame of per-env items only: the in-kernel path handles the empty gather)
    pl->num_units = units;
    pl->num_owners = owners;
}
}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_observe(const GfObservationArgs* a, void* stream) {
    uint32_t needs = 0;
    int vec = 1;
    const int rc = gf::observe_prep(a, &needs, &vec);
    if (rc) return rc;
    if (a->num_envs == 0) return GF_OK;
    gf::ObsKernelArgs k;
    k.a = *a;
    k.needs = needs;
    k._pad = 0;
    gf::observe_plan(a, &k.plan);
    const bool vec4 = vec == 4, vec2 = vec == 2;
    const int O = a->obs_width;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)(O + 1) * gf::kEnvBlock * sizeof(float);
    gf::PhaseScope scope(GF_PHASE_OBSERVE, s);
    scope.begin_bracket();
    const unsigned grid = gf::env_grid(a->num_envs);  // one 256-thread workgroup per 64-env tile
    if (vec4) gf::klaunch(gf::observe_kernel<4>, dim3(grid), dim3(gf::kObsBlock), lds, s, k);
    else if (vec2) gf::klaunch(gf::observe_kernel<2>, dim3(grid), dim3(gf::kObsBlock), lds, s, k);
    else gf::klaunch(gf::observe_kernel<1>, dim3(grid), dim3(gf::kObsBlock), lds, s, k);
    return gf::launch_status();
}
#endif
