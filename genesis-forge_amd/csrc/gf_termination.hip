// gf_termination.hip — Phase B1+B3: TerminationManager.step as one launch.
//
// Replaces managers/termination_manager.py:151-190 and the term bodies of mdp/terminations.py
// (≈27 aten ops, 20 launches and one nonzero() host sync per term in the reference).
//
// One lane per env.  All state a term can need (quat, episode_length, max_episode_length, pos) is
// requested up front so every load of the wave is in flight at once; the term table is read from
// the kernarg segment with wave-uniform (scalar) loads, so the per-term switch never diverges.
// Per-term "fired" counts for the "Terminations / <name>" log (termination_manager.py:178-182) are a
// wave ballot + one integer atomic per wave that has a hit (integer adds: order independent).
// Algorithmic traffic (K=2: timeout + bad_orientation): R quat 16 + ep_len 4 + max_len 4, W 2 masks.
#include "gf_launch.h"
#include "gf_terms.h"

namespace gf {

enum : uint32_t { NEED_QUAT = 1, NEED_POS = 2, NEED_EPLEN = 4, NEED_MAXLEN = 8 };

// the phase for the 64 envs of workgroup blockIdx.x; lanes = threadIdx.x < 64 (also called from the phase-chain kernels)
__device__ __forceinline__ void termination_body(const GfTerminationArgs& a, const uint32_t needs) {
    const int64_t n = (int64_t)blockIdx.x * kEnvBlock + threadIdx.x;
    const bool live = n < a.num_envs;
    const int64_t m = live ? n : 0;  // clamp so every lane can take part in ballots

    float4 q = make_float4(1.f, 0.f, 0.f, 0.f);
    V3 pos{0.f, 0.f, 0.f};
    int ep_len = 0, max_len = 0;
    if (needs & NEED_QUAT) q = load_quat(a.entity.quat, m);
    if (needs & NEED_POS) pos = load3(a.entity.pos, m);
    if (needs & NEED_EPLEN) ep_len = a.episode_length[m];
    if (needs & NEED_MAXLEN) max_len = a.max_episode_length[m];

    // projected gravity is shared by every bad_orientation term
    float tilt_sin = 0.f;
    if (needs & NEED_QUAT) {
        const V3 g = rot_inv(q, V3{0.f, 0.f, -1.f});
        tilt_sin = clamp_max(norm2(g.x, g.y), 0.99f);  // torch.clamp(max=0.99), NaN propagates
    }

    TermRegs tr;
    tr.ep_len = ep_len; tr.max_len = max_len; tr.has_maxlen = (needs & NEED_MAXLEN) != 0; tr.tilt_sin = tilt_sin; tr.pos = pos; tr.m = m;
    int term = 0, trunc = 0;
    const int K = a.num_terms;
    for (int k = 0; k < K; ++k) {
        const GfTerm& t = a.terms[k];
        int v = 0;
        v = eval_termination_term(t, a, tr, needs & NEED_MAXLEN);
        v = live ? v : 0;
        if (t.flags & GF_TERM_FLAG_TIME_OUT) trunc |= v; else term |= v;
        if (a.term_out && live) a.term_out[(int64_t)k * a.num_envs + n] = (uint8_t)v;
        if (a.stats) {
            const unsigned long long hit = __ballot(v);
            if (hit && threadIdx.x == 0) atomicAdd(&stats_shard(a.stats)->term_fired[k], popc64(hit));
        }
    }
    if (live) {
        a.terminated[n] = (uint8_t)term;
        a.truncated[n] = (uint8_t)trunc;
    }
}

#ifndef GF_BODIES_ONLY
__global__ __launch_bounds__(kEnvBlock) void termination_kernel(const GfTerminationArgs a, const uint32_t needs) {
    prefetch_args<GfTerminationArgs>();
    termination_body(a, needs);
}
#endif

}  // namespace gf

#ifndef GF_BODIES_ONLY
namespace gf {
// validation + which inputs the term table needs (shared by the entry point and the phase-chain launcher)
int termination_prep(const GfTerminationArgs* a, uint32_t* needs_out) {
    if (!a || !a->terminated || !a->truncated) return GF_E_NULL;
    if (a->num_terms < 0 || a->num_terms > GF_MAX_TERM_TERMS || a->num_envs < 0) return GF_E_RANGE;
    uint32_t needs = 0;
    for (int k = 0; k < a->num_terms; ++k) {
        const GfTerm& t = a->terms[k];
        switch (t.op) {
            case GF_T_TIMEOUT:
                if (a->max_episode_length) needs |= gf::NEED_EPLEN | gf::NEED_MAXLEN;
                break;
            case GF_T_BAD_ORIENTATION: needs |= gf::NEED_QUAT | gf::NEED_EPLEN; break;
            case GF_T_BASE_HEIGHT_BELOW:
            case GF_T_OUT_OF_BOUNDS: needs |= gf::NEED_POS; break;
            case GF_T_CONTACT_FORCE_GRACE: needs |= gf::NEED_EPLEN;  // fallthrough
            case GF_T_HAS_CONTACT:
            case GF_T_CONTACT_FORCE:
                if (t.i[0] < 0 || t.i[0] >= GF_MAX_CONTACT_VIEWS || !a->contact[t.i[0]].contacts) return GF_E_SLOT;
                if (a->contact[t.i[0]].num_links <= 0) return GF_E_RANGE;
                break;
            case GF_T_EXTERNAL:
                if (t.i[0] < 0 || t.i[0] >= GF_MAX_EXT || !a->ext[t.i[0]]) return GF_E_SLOT;
                break;
            default: return GF_E_OPCODE;
        }
    }
    if ((needs & gf::NEED_QUAT) && !a->entity.quat) return GF_E_NULL;
    if ((needs & gf::NEED_POS) && !a->entity.pos) return GF_E_NULL;
    if ((needs & gf::NEED_EPLEN) && !a->episode_length) return GF_E_NULL;
    *needs_out = needs;
    return GF_OK;
}
}  // namespace gf

extern "C" __attribute__((visibility("default"))) int gf_termination_step(const GfTerminationArgs* a, void* stream) {
    uint32_t needs = 0;
    const int rc = gf::termination_prep(a, &needs);
    if (rc) return rc;
    if (a->num_envs == 0) return GF_OK;
    hipStream_t s = (hipStream_t)stream;
    gf::PhaseScope scope(GF_PHASE_TERMINATION, s);
    scope.begin_bracket();
    gf::klaunch(gf::termination_kernel, dim3(gf::env_grid(a->num_envs)), dim3(gf::kEnvBlock), 0, s, *a, needs);
    return gf::launch_status();
}
#endif
