// gf_api.hip — library-level entry points: version, error strings, stats clear, opt-in
// HIP-event profiling around the kernel of one phase (used by bench.py for roofline.achieved).
#include <cstdlib>
#include <vector>

#include "gf_launch.h"

namespace gf {
// (GF_NO_CONTACT_FOLD / GF_FORCE_CONTACT_FOLD: the measurement switches of tools/scaling_table.sh — never / always, against the size policy)
int g_options[GF_OPT_COUNT] = {2, 0, 0, 1, getenv("GF_NO_CONTACT_FOLD") ? 0 : (getenv("GF_FORCE_CONTACT_FOLD") ? 2 : 1)};
Profiler g_prof;
thread_local LaunchSink g_sink;
bool contact_compatible(const GfContactArgs* x, const GfContactArgs* y);          // gf_contact.hip
int contact_launch(const GfContactArgs* const* mgrs, int num, hipStream_t s);
int post_step(const GfPostRefs* r, const GfContactArgs* const* mgrs, int num_mgr, hipStream_t s);   // gf_post.hip
int chain_a_try(const GfOp* ops, int i, int num_ops, hipStream_t s, int* rc, DeferredFlags* deferred);   // gf_chain.hip
int chain_b_try(const GfOp* ops, int i, int num_ops, hipStream_t s, int* rc, DeferredFlags* deferred);
int gait_launch(const GfGaitArgs* a, hipStream_t s, bool flags_all);                                      // gf_gait.hip
int unroll_pair(const GfHistoryUnrollArgs* a, const GfHistoryUnrollArgs* b, hipStream_t s, int* fused);   // gf_unroll.hip
}

#define GF_EXPORT __attribute__((visibility("default")))
namespace gf {
__global__ __launch_bounds__(256) void stats_pack_kernel(const GfStatsPackArgs a) { fold_stats_block256(a.src, a.dst, nullptr); }

// One wave: lane r looks at row r's reset count, the ballot names the newest row that reset something, the lanes copy it.
__global__ __launch_bounds__(GF_WAVE) void stats_last_reset_kernel(const double* rows, const int num_rows, double* dst) {
    const int lane = (int)threadIdx.x;
    const bool hit = lane < num_rows && rows[(size_t)lane * GF_STATS_VECTOR_LEN + GF_MAX_TERM_TERMS] > 0.0;
    const unsigned long long m = __ballot(hit);
    if (!m) return;
    const int r = 63 - __builtin_clzll(m);
    for (int v = lane; v < GF_STATS_VECTOR_LEN; v += GF_WAVE) dst[v] = rows[(size_t)r * GF_STATS_VECTOR_LEN + v];
}
}  // namespace gf

extern "C" {

GF_EXPORT int gf_abi_version(void) { return GF_ABI_VERSION; }

GF_EXPORT int gf_set_option(int option, int value) {
    if (option < 0 || option >= GF_OPT_COUNT) return GF_E_RANGE;
    gf::g_options[option] = value;
    return GF_OK;
}

GF_EXPORT int gf_sizeof(int which) {
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

GF_EXPORT const char* gf_build_info(void) { return "genesis-forge_amd gf_step: gfx950 HIP, -ffp-contract=off, built " __DATE__ " " __TIME__; }

GF_EXPORT const char* gf_error_string(int code) {
    switch (code) {
        case GF_OK: return "ok";
        case GF_E_NULL: return "required pointer is NULL";
        case GF_E_RANGE: return "size or count out of supported range";
        case GF_E_OPCODE: return "unknown opcode in term table";
        case GF_E_SLOT: return "term references an unbound view/slot";
        case GF_E_UNSUPPORTED: return "unsupported configuration";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

GF_EXPORT int gf_stats_clear(GfStepStats* stats, void* stream) {
    if (!stats) return GF_E_NULL;
    GF_HIP_CHECK(hipMemsetAsync(stats, 0, sizeof(GfStepStats) * GF_STATS_SHARDS, (hipStream_t)stream));
    return GF_OK;
}

GF_EXPORT int gf_stats_pack(const GfStatsPackArgs* a, void* stream) {
    if (!a || !a->src || !a->dst) return GF_E_NULL;
    gf::klaunch(gf::stats_pack_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, *a);
    return gf::launch_status();
}

GF_EXPORT int gf_stats_last_reset(const double* rows, int num_rows, double* dst, void* stream) {
    if (!rows || !dst) return GF_E_NULL;
    if (num_rows < 0 || num_rows > GF_WAVE) return GF_E_RANGE;
    if (num_rows == 0) return GF_OK;
    gf::klaunch(gf::stats_last_reset_kernel, dim3(1), dim3(GF_WAVE), 0, (hipStream_t)stream, rows, num_rows, dst);
    return gf::launch_status();
}

// Replay a recorded step as ONE hipGraphLaunch.  *cache is an opaque handle owned by the caller (NULL at first).  The ops run
// through their ordinary entry points — validation, packing, kernel selection — but their launches land in the launch sink
// (gf_launch.h): the first call builds and instantiates a linear graph of kernel nodes, later calls refresh each node's
// arguments in place and launch the graph; if the launch sequence changed shape (another kernel variant, another grid) the
// graph is rebuilt.  Steps that cannot be expressed as kernel nodes only (memset / copy ops, a profiled phase, the option
// switched off) run through gf_run_ops unchanged.
static bool graphable(const GfOp* ops, int num_ops) {
    if (!gf::g_options[GF_OPT_GRAPH] || gf::g_prof.phase >= 0) return false;
    for (int i = 0; i < num_ops; ++i)
        if (ops[i].phase == GF_OP_STATS_CLEAR || ops[i].phase == GF_OP_STATS_COPY) return false;
    return true;
}

static void graph_free(gf::GraphCache* g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

GF_EXPORT int gf_graph_destroy(void** cache) {
    if (cache && *cache) { graph_free((gf::GraphCache*)*cache); *cache = nullptr; }
    return GF_OK;
}

GF_EXPORT int gf_run_ops_graph(void** cache, const GfOp* ops, int num_ops, void* stream, int* failed_index) {
    if (!cache) return GF_E_NULL;
    if (!graphable(ops, num_ops)) return gf_run_ops(ops, num_ops, stream, failed_index);
    gf::LaunchSink& k = gf::g_sink;
    gf::GraphCache* g = (gf::GraphCache*)*cache;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool build = g == nullptr;
        if (build) {
            g = new gf::GraphCache();
            if (hipGraphCreate(&g->graph, 0) != hipSuccess) { graph_free(g); *cache = nullptr; return gf_run_ops(ops, num_ops, stream, failed_index); }
        }
        k = gf::LaunchSink();
        k.mode = build ? gf::SINK_BUILD : gf::SINK_UPDATE;
        k.g = g;
        const int rc = gf_run_ops(ops, num_ops, stream, failed_index);
        const bool mismatch = k.mismatch || (!build && k.cursor != g->nodes.size());
        const hipError_t err = k.error;
        k = gf::LaunchSink();  // back to direct launches whatever happened
        if (rc != GF_OK || err != hipSuccess) {   // an op failed validation, or the graph API refused: drop the graph
            graph_free(g);
            *cache = nullptr;
            if (rc != GF_OK) return rc;
            return gf_run_ops(ops, num_ops, stream, failed_index);
        }
        if (build) {
            if (g->nodes.empty() || hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0) != hipSuccess) {
                graph_free(g);
                *cache = nullptr;
                return gf_run_ops(ops, num_ops, stream, failed_index);
            }
            *cache = g;
        } else if (mismatch) {   // the step no longer has the recorded shape: rebuild once
            graph_free(g);
            g = nullptr;
            *cache = nullptr;
            continue;
        }
        const hipError_t e = hipGraphLaunch(g->exec, (hipStream_t)stream);
        if (e == hipSuccess) return GF_OK;
        // the runtime refuses graph launches here: plain launches from now on (nothing of this step has been enqueued yet)
        (void)hipGetLastError();
        graph_free(g);
        *cache = nullptr;
        gf::g_options[GF_OPT_GRAPH] = 0;
        return gf_run_ops(ops, num_ops, stream, failed_index);
    }
    return gf_run_ops(ops, num_ops, stream, failed_index);
}


/* the patch table of a recorded step (gf_step.h: GfReplay) */
}  // extern "C"
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
extern "C" {

GF_EXPORT int gf_replay_step(const GfReplay* r, const void* actions, const void* const* params, int num_params, void* stream, int* failed_index) {
    if (failed_index) *failed_index = -1;
    const int rc = replay_patch(r, actions, params, num_params);
    if (rc != GF_OK || !r->ops || r->num_ops <= 0) return rc;
    return gf_run_ops(r->ops, r->num_ops, stream, failed_index);
}

GF_EXPORT int gf_run_ops(const GfOp* ops, int num_ops, void* stream, int* failed_index) {
    if (!ops || num_ops < 0) return GF_E_NULL;
    hipStream_t s = (hipStream_t)stream;
    gf::DeferredFlags deferred;
    for (int i = 0; i < num_ops; ++i) {
        int rc = GF_OK;
        const void* a = ops[i].args;
        switch (ops[i].phase) {
            case GF_PHASE_ACTION: rc = gf_action_step((const GfActionArgs*)a, stream); break;
            case GF_PHASE_CONTACT: {
                // consecutive ContactManagers over the same scene arrays share one launch (slot ids read once for all of them)
                const GfContactArgs* run[4] = {(const GfContactArgs*)a};
                int cnt = 1, links = run[0] ? run[0]->num_targets : 0;
                while (run[0] && cnt < 4 && i + cnt < num_ops && ops[i + cnt].phase == GF_PHASE_CONTACT && ops[i + cnt].args) {
                    const GfContactArgs* nx = (const GfContactArgs*)ops[i + cnt].args;
                    if (!gf::contact_compatible(run[0], nx) || links + nx->num_targets > 64) break;
                    links += nx->num_targets;
                    run[cnt++] = nx;
                }
                // … and when the fused post-physics launch follows them, its first phase (the same 64-env tiles, one launch less)
                if (run[0] && i + cnt < num_ops && ops[i + cnt].phase == GF_OP_POST_PHYSICS && ops[i + cnt].args) {
                    rc = gf::post_step((const GfPostRefs*)ops[i + cnt].args, run, cnt, s);
                    if (rc == GF_OK) { i += cnt; break; }
                    if (rc != GF_E_UNSUPPORTED) { i += cnt; break; }   // (a real failure belongs to the post-physics op)
                }
                rc = gf::contact_launch(run, cnt, s);
                if (rc == GF_OK) i += cnt - 1;
            } break;
            case GF_PHASE_TERMINATION: {
                const int used = gf::chain_a_try(ops, i, num_ops, s, &rc, &deferred);   // termination → reward → command.step … in one launch
                if (used > 0) { if (rc == GF_OK) i += used - 1; break; }
                rc = gf_termination_step((const GfTerminationArgs*)a, stream);
            } break;
            case GF_PHASE_REWARD: {
                const int used = gf::chain_a_try(ops, i, num_ops, s, &rc, &deferred);   // reward → command.step … (no termination op in front)
                if (used > 0) { if (rc == GF_OK) i += used - 1; break; }
                rc = gf_reward_step((const GfRewardArgs*)a, stream);
            } break;
            case GF_PHASE_COMMAND: rc = gf_command_step((const GfCommandArgs*)a, stream); break;
            case GF_PHASE_RESET: {
                const int used = gf::chain_b_try(ops, i, num_ops, s, &rc, &deferred);   // reset → command.reset … → observe … in one launch
                if (used > 0) { if (rc == GF_OK) i += used - 1; break; }
                rc = gf_masked_reset((const GfResetArgs*)a, stream);
            } break;
            case GF_PHASE_OBSERVE: rc = gf_observe((const GfObservationArgs*)a, stream); break;
            case GF_PHASE_ROTATE: rc = gf_entity_rotate((const GfRotateArgs*)a, stream); break;
            case GF_PHASE_SCENE: rc = gf_synth_scene_step((const GfSynthSceneArgs*)a, stream); break;
            case GF_PHASE_TERRAIN: rc = gf_terrain_height((const GfTerrainHeightArgs*)a, stream); break;
            case GF_PHASE_ROLLOUT: rc = gf_rollout_write((const GfRolloutArgs*)a, stream); break;
            case GF_PHASE_ROLLOUT_POLICY: rc = gf_rollout_policy_write((const GfRolloutPolicyArgs*)a, stream); break;
            case GF_PHASE_GAE: rc = gf_gae((const GfGaeArgs*)a, stream); break;
            case GF_PHASE_COMPACT: rc = gf_done_compact((const GfCompactArgs*)a, stream); break;
            case GF_PHASE_UNROLL: {   // the gathers of two managers (policy + critic) share a launch
                int fused = 0;
                if (a && i + 1 < num_ops && ops[i + 1].phase == GF_PHASE_UNROLL && ops[i + 1].args)
                    rc = gf::unroll_pair((const GfHistoryUnrollArgs*)a, (const GfHistoryUnrollArgs*)ops[i + 1].args, s, &fused);
                if (fused) { if (rc == GF_OK) i += 1; break; }
                if (rc == GF_OK) rc = gf_history_unroll((const GfHistoryUnrollArgs*)a, stream);
            } break;
            case GF_PHASE_GAIT: {
                const GfGaitArgs* g = (const GfGaitArgs*)a;
                const bool all = g && g->mode != GF_CMD_STEP && deferred.has(g->state);
                rc = gf::gait_launch(g, s, all);
                if (all) deferred.pop(g->state);
            } break;
            case GF_OP_STATS_CLEAR: rc = gf_stats_clear((GfStepStats*)const_cast<void*>(a), stream); break;
            case GF_OP_POST_PHYSICS: rc = gf_post_physics_step((const GfPostRefs*)a, stream); break;
            case GF_OP_STATS_PACK: rc = gf_stats_pack((const GfStatsPackArgs*)a, stream); break;
            case GF_OP_STATS_COPY: {
                const GfStatsCopyArgs* c = (const GfStatsCopyArgs*)a;
                if (!c || !c->src || !c->dst) { rc = GF_E_NULL; break; }
                hipError_t e = hipMemcpyAsync(c->dst, c->src, sizeof(GfStepStats) * GF_STATS_SHARDS, hipMemcpyDeviceToHost, s);
                if (e == hipSuccess && c->event) e = hipEventRecord((hipEvent_t)c->event, s);
                rc = (int)e;
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

GF_EXPORT void* gf_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return (void*)e;
}

GF_EXPORT int gf_event_destroy(void* event) { return event ? (int)hipEventDestroy((hipEvent_t)event) : GF_OK; }

GF_EXPORT int gf_event_synchronize(void* event) { return event ? (int)hipEventSynchronize((hipEvent_t)event) : GF_OK; }

GF_EXPORT int gf_profile_begin(int phase, int max_samples) {
    if (phase < 0 || phase >= GF_PHASE_COUNT || max_samples <= 0) return GF_E_RANGE;
    gf::Profiler& p = gf::g_prof;
    for (hipEvent_t e : p.events) hipEventDestroy(e);
    p.events.clear();
    p.events.resize((size_t)max_samples * 2);
    for (auto& e : p.events) GF_HIP_CHECK(hipEventCreate(&e));
    p.count = 0;
    p.calls = 0;
    p.max_samples = max_samples;
    p.phase = phase;
    return GF_OK;
}

GF_EXPORT int gf_profile_end(double* total_ms, int* samples) {
    gf::Profiler& p = gf::g_prof;
    double tot = 0.0;
    for (int i = 0; i < p.count; ++i) {
        GF_HIP_CHECK(hipEventSynchronize(p.events[2 * i + 1]));
        float ms = 0.f;
        GF_HIP_CHECK(hipEventElapsedTime(&ms, p.events[2 * i], p.events[2 * i + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (samples) *samples = p.count;
    for (hipEvent_t e : p.events) hipEventDestroy(e);
    p.events.clear();
    p.phase = -1;
    p.count = 0;
    p.max_samples = 0;
    return GF_OK;
}

}  // extern "C"
