// gf_unroll.hip — in-place history ring → the reference's newest-first observation tensor, one streaming launch.
//
// Replaces `torch.cat(self._history, dim=-1)` of observation_manager.py:226 for managers whose history is kept as a ring
// (GfObservationArgs.history_ring): out[n, j·O + c] = ring[n, (k + j) mod H, c], k = slot of the newest frame.
//
// Why a launch of its own (measured): shifting the previous output inside the fused post-physics kernel is a chain of
// load-batch → wait (for the loads AND the previous batch's stores: one vmcnt on gfx9) → store-batch run by the 12 waves per CU a
// 64-env tile layout leaves — 3.4 TB/s on the gait task's 184 MB; history workgroups inside that launch (tried) write the same cache
// lines as the tile workgroups from another XCD at another time and are slower still.  This copy depends on nothing but the ring:
// 2 048-float chunks, 16-byte units aligned on `out` (rows of an odd-width frame are not 16-byte aligned, the array is), two units
// per lane in flight, thousands of workgroups — tools/microbench_shift.hip puts the structure at 6.7 TB/s.
// A unit that lies inside one frame is one dword-aligned 16-byte load; the ≤ 1 unit per frame that straddles a frame edge (O mod 4 ≠ 0)
// is two of them and a select.  Algorithmic traffic: R 4·O·H + W 4·O·H bytes per env.
#include "gf_launch.h"
#include "gf_obs_hist.h"   // f32x4u / f32x4a

namespace gf {

constexpr int kUnrollBlock = 256;
#ifndef GF_UNROLL_UNITS   // (A/B builds: tools/ab_build.sh)
#define GF_UNROLL_UNITS 2
#endif
constexpr int kUnrollUnits = GF_UNROLL_UNITS;   // 16-byte units per lane in flight (2 / 4 / 8 measured on one box, whole gait step at 65 536 envs: 127.1 / 128.2 / 134.3 µs — profiles/r03_w_ab_unroll.jsonl)
constexpr int kUnrollChunk = kUnrollBlock * kUnrollUnits * 4;   // floats per workgroup

// Division by the two run-time widths (O·H per row, O per frame) as multiply-shift with magic numbers the HOST computes once per
// launch: in the kernel a 64-bit division is a software loop of several hundred cycles per wave — a tenth of such a wave's life.
//   i / d = (i · ceil(2^40 / d)) >> 40 for i < 2^40 / d (d < 2^17, i < 2^18 here);  e / d = umulhi64(e, ceil(2^64 / d)) for e·d < 2^64.
struct UnrollConsts {
    uint64_t m_oh, m_o;   // ceil(2^40 / (O·H)), ceil(2^40 / O)
    uint64_t big_oh;      // ceil(2^64 / (O·H)); 0 when O·H == 1 (then e / 1 = e)
};

struct UnrollMap {
    uint64_t m_oh, m_o;
    int O, OH, H, head;
    __device__ __forceinline__ UnrollMap(const UnrollConsts& c, int O_, int H_, int head_) : m_oh(c.m_oh), m_o(c.m_o), O(O_), OH(O_ * H_), H(H_), head(head_) {}
    // element `ec` floats after the start of a row (it may lie in a later row) → offset of its source in the ring, from the same row start;
    // `cc` = its column inside its frame
    __device__ __forceinline__ int src(int ec, int& cc) const {
        const int dn = (int)(((uint64_t)(uint32_t)ec * m_oh) >> 40), c = ec - dn * OH;
        const int j = (int)(((uint64_t)(uint32_t)c * m_o) >> 40);
        cc = c - j * O;
        int s = head + j;
        s = s >= H ? s - H : s;
        return dn * OH + s * O + cc;
    }
};
__device__ __forceinline__ int64_t rows_before(int64_t e, const UnrollConsts& c) { return c.big_oh ? (int64_t)__umul64hi((uint64_t)e, c.big_oh) : e; }

// Units across a frame edge (O mod 4 != 0: one unit in ~O/4; also the unit across a row edge).  A frame is at least four floats wide
// here, so such a unit is [the last a floats of one frame | the first 4 - a floats of the next]: TWO 16-byte loads that stay inside
// their frames — the four floats that END the first frame, the four that START the second — and a select.  Every load of a lane
// (whole units and edge halves) is issued before the one wait; a first version composed edge units from element loads AFTER the
// stores, and since every wave meets an edge that put four more serialised round trips on every wave: 48 us instead of 25 for the
// gait policy history at 65 536 envs (profiles/r02_m_unroll_edges.txt).
// Streaming (`NT`): a gather whose ring + output are larger than the caches can hold from one step to the next (kStreamBytes) reads
// and writes with the non-temporal hint — nothing of it is found again by the next launch anyway, and without the hint the 200 MB of
// the gait policy history at 65 536 envs pushed every OTHER kernel's working set out of L2 / MALL each step: the whole gait step
// went from 129.6 to 114.8 us with it, the same step at 8 192 envs (12 MB ring, re-read from cache every step) from 35.7 to 36.7
// the other way — hence the size rule (profiles/r03_z_ab_nt.jsonl; the threshold at 16 MB or 0 instead: 51 MB of gather at 16 384 envs
// 45.3 -> 47.1 us, 38 MB at 12 288 envs 40.1 -> 41.6, r03_z_ab_gather_units_threshold.jsonl — which also re-checks the units per lane
// with the hint on: 1 / 2 / 4 / 8 = 112.7 / 112.5 / 114.8 / 119.1 us).
#ifndef GF_STREAM_MB   // (A/B builds)
#define GF_STREAM_MB 64
#endif
constexpr int64_t kStreamBytes = (int64_t)GF_STREAM_MB << 20;
template <bool NT, class T>
__device__ __forceinline__ T stream_load(const GF_GLOBAL T* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, class T>
__device__ __forceinline__ void stream_store(GF_GLOBAL T* p, const T v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT>
__device__ __forceinline__ void unroll_chunk(const GfHistoryUnrollArgs& a, const UnrollConsts& uc, const unsigned block) {
    const UnrollMap map(uc, a.frame_width, a.history_len, a.ring_slot - 1);
    const int64_t total = a.num_envs * (int64_t)map.OH, units = total >> 2;
    const int64_t e0 = (int64_t)block * kUnrollChunk;
    const int64_t n0 = rows_before(e0, uc);   // workgroup-uniform
    const int c0 = (int)(e0 - n0 * map.OH);
    const GF_GLOBAL float* ring = G(a.ring) + n0 * map.OH;
    GF_GLOBAL float* out = G(a.out) + n0 * map.OH;
    GF_GLOBAL float* out2 = a.out2 ? G(a.out2) + n0 * map.OH : nullptr;
    const int tid = (int)threadIdx.x;
    const bool wide = map.O >= 4;   // uniform
    const GF_GLOBAL float* vbase = wide ? ring : (const GF_GLOBAL float*)g_zero_pad;
    f32x4u v[kUnrollUnits], w[kUnrollUnits];
    int at[kUnrollUnits], lead[kUnrollUnits];   // lead: floats of the unit that belong to the frame its first float is in (>= 4: all)
    uint32_t on_mask = 0u;
#pragma unroll
    for (int k = 0; k < kUnrollUnits; ++k) {
        const int lu = tid + k * kUnrollBlock;
        const bool on = (e0 >> 2) + lu < units;
        const int ec = c0 + (on ? lu << 2 : 0);
        int cc;
        const int so = map.src(ec, cc);
        const int la = map.O - cc;
        at[k] = ec;
        lead[k] = on ? la : 4;
        on_mask |= on ? 1u << k : 0u;
        // unconditional, back to back.  A lane past the array reads `ring + 0` — 16 bytes that exist when a frame is at least a unit
        // wide; with narrower frames (`!wide`: the element path below does the loading) `ring + 0` may end less than 16 bytes before
        // the end of the tensor (O*H < 4, last row), so the wave-uniform base is the code object's zero pad then
        v[k] = stream_load<NT>(reinterpret_cast<const GF_GLOBAL f32x4u*>(vbase + (!on || !wide ? 0 : (la >= 4 ? so : so + la - 4))));
        w[k] = f32x4u{0.f, 0.f, 0.f, 0.f};   // (not v[k]: a copy would wait for the load)
    }
    if (wide) {
#pragma unroll
        for (int k = 0; k < kUnrollUnits; ++k)
            if (lead[k] < 4) {   // the unit's second frame: its first four floats
                int cc;
                w[k] = stream_load<NT>(reinterpret_cast<const GF_GLOBAL f32x4u*>(ring + map.src(at[k] + lead[k], cc)));
            }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), once
#pragma unroll
        for (int k = 0; k < kUnrollUnits; ++k)
            if ((on_mask >> k) & 1u) {
                const int la = lead[k];
                f32x4a r{v[k].x, v[k].y, v[k].z, v[k].w};
                if (la == 1) r = f32x4a{v[k].w, w[k].x, w[k].y, w[k].z};
                else if (la == 2) r = f32x4a{v[k].z, v[k].w, w[k].x, w[k].y};
                else if (la == 3) r = f32x4a{v[k].y, v[k].z, v[k].w, w[k].x};
                stream_store<NT>(reinterpret_cast<GF_GLOBAL f32x4a*>(out + at[k]), r);
                if (out2) stream_store<NT>(reinterpret_cast<GF_GLOBAL f32x4a*>(out2 + at[k]), r);
            }
    } else {   // frames narrower than a unit: element by element
#pragma unroll
        for (int k = 0; k < kUnrollUnits; ++k)
            if ((on_mask >> k) & 1u) {
                int cc;
                f32x4a r;
                r.x = ring[map.src(at[k], cc)];
                r.y = ring[map.src(at[k] + 1, cc)];
                r.z = ring[map.src(at[k] + 2, cc)];
                r.w = ring[map.src(at[k] + 3, cc)];
                *reinterpret_cast<GF_GLOBAL f32x4a*>(out + at[k]) = r;
                if (out2) *reinterpret_cast<GF_GLOBAL f32x4a*>(out2 + at[k]) = r;
            }
    }
    // the array's last total mod 4 floats (they belong to the last row)
    const int tail = (int)(total & 3);
    if (block == 0 && tid < tail) {
        const int64_t e = (units << 2) + tid, n = rows_before(e, uc);
        int cc;
        const int so = map.src((int)(e - n * map.OH), cc);
        const float r = G(a.ring)[n * map.OH + so];
        G(a.out)[e] = r;
        if (a.out2) G(a.out2)[e] = r;
    }
}

template <bool NT>
__global__ __launch_bounds__(kUnrollBlock) void history_unroll_kernel(const GfHistoryUnrollArgs a, const UnrollConsts uc) { unroll_chunk<NT>(a, uc, blockIdx.x); }

// two managers (policy + critic of one env), one launch: workgroups [0, split) gather the first, the rest the second
template <bool NT>
__global__ __launch_bounds__(kUnrollBlock) void history_unroll2_kernel(const GfHistoryUnrollArgs a, const UnrollConsts ua, const GfHistoryUnrollArgs b,
                                                                        const UnrollConsts ub, const unsigned split) {
    if (blockIdx.x < split) unroll_chunk<NT>(a, ua, blockIdx.x);
    else unroll_chunk<NT>(b, ub, blockIdx.x - split);
}

int unroll_prep(const GfHistoryUnrollArgs* a) {
    if (!a) return GF_E_NULL;
    if (a->num_envs < 0 || a->frame_width < 1 || a->history_len < 1 || a->ring_slot < 1 || a->ring_slot > a->history_len) return GF_E_RANGE;
    if ((int64_t)a->frame_width * a->history_len >= (1 << 17)) return GF_E_RANGE;   // FastDiv's exact range
    if (a->num_envs == 0) return GF_OK;
    if (!a->ring || !a->out) return GF_E_NULL;
    if ((reinterpret_cast<uintptr_t>(a->out) & 15u) || (reinterpret_cast<uintptr_t>(a->out2) & 15u) || (reinterpret_cast<uintptr_t>(a->ring) & 3u)) return GF_E_UNSUPPORTED;
    return GF_OK;
}



static UnrollConsts unroll_consts(const GfHistoryUnrollArgs* a) {
    const uint64_t oh = (uint64_t)a->frame_width * (uint64_t)a->history_len, o = (uint64_t)a->frame_width;
    UnrollConsts uc;
    uc.m_oh = (((uint64_t)1 << 40) + oh - 1) / oh;
    uc.m_o = (((uint64_t)1 << 40) + o - 1) / o;
    uc.big_oh = oh > 1 ? ~(uint64_t)0 / oh + 1 : 0;   // ceil(2^64 / oh) for oh >= 2 (2^64 is a multiple of oh only for powers of two, where this is exact too)
    return uc;
}
static int64_t unroll_blocks(const GfHistoryUnrollArgs* a) {
    const int64_t total = a->num_envs * (int64_t)a->frame_width * a->history_len;
    return (total + kUnrollChunk - 1) / kUnrollChunk;
}

// ring + output (+ second output) of the launch, in bytes, against kStreamBytes
static bool unroll_streams(const GfHistoryUnrollArgs* a, const GfHistoryUnrollArgs* b) {
    auto bytes = [](const GfHistoryUnrollArgs* x) { return x ? x->num_envs * (int64_t)x->frame_width * x->history_len * 4 * (x->out2 ? 3 : 2) : (int64_t)0; };
    return bytes(a) + bytes(b) >= kStreamBytes;
}

// gf_run_ops: two consecutive gather ops share a launch.  Returns GF_OK or an error; *fused = 1 when both were launched.
int unroll_pair(const GfHistoryUnrollArgs* a, const GfHistoryUnrollArgs* b, hipStream_t s, int* fused) {
    *fused = 0;
    int rc = unroll_prep(a);
    if (rc) return rc;
    if (unroll_prep(b) != GF_OK || a->num_envs == 0 || b->num_envs == 0 || g_prof.phase == GF_PHASE_UNROLL) return GF_OK;   // caller launches them one by one
    const int64_t na = unroll_blocks(a), nb = unroll_blocks(b);
    if (na + nb >= (int64_t)1 << 31) return GF_OK;
    if (unroll_streams(a, b)) klaunch(history_unroll2_kernel<true>, dim3((unsigned)(na + nb)), dim3(kUnrollBlock), 0, s, *a, unroll_consts(a), *b, unroll_consts(b), (unsigned)na);
    else klaunch(history_unroll2_kernel<false>, dim3((unsigned)(na + nb)), dim3(kUnrollBlock), 0, s, *a, unroll_consts(a), *b, unroll_consts(b), (unsigned)na);
    *fused = 1;
    return launch_status();
}

}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_history_unroll(const GfHistoryUnrollArgs* a, void* stream) {
    const int rc = gf::unroll_prep(a);
    if (rc) return rc;
    if (a->num_envs == 0) return GF_OK;
    const int64_t blocks = gf::unroll_blocks(a);
    if (blocks >= (int64_t)1 << 31) return GF_E_RANGE;
    hipStream_t s = (hipStream_t)stream;
    gf::PhaseScope scope(GF_PHASE_UNROLL, s);
    if (gf::unroll_streams(a, nullptr)) GF_LAUNCH(scope, gf::history_unroll_kernel<true>, (unsigned)blocks, gf::kUnrollBlock, 0, s, *a, gf::unroll_consts(a));
    else GF_LAUNCH(scope, gf::history_unroll_kernel<false>, (unsigned)blocks, gf::kUnrollBlock, 0, s, *a, gf::unroll_consts(a));
    return gf::launch_status();
}
