// gf_launch.h — host-side launch helpers: grid sizing and the opt-in per-phase event profiler.
#pragma once

#include <hip/hip_ext.h>

#include <vector>

#include "gf_device.h"
#include "gf_prefetch.h"

namespace gf {

// One lane per env, one wave (64 lanes) per workgroup: N/64 workgroups spread over the 256 CUs /
// 8 XCDs round-robin.  These kernels stream every byte exactly once and share nothing between
// workgroups, so there is no L2 reuse for an XCD-aware tile map to protect; what matters is that
// even N=4096 still puts a wave on 64 different CUs and N=65536 gives every CU 4 waves with all of
// their loads in flight at once.
constexpr int kEnvBlock = 64;

inline unsigned env_grid(int64_t n, int block = kEnvBlock) { return (unsigned)((n + block - 1) / block); }

struct Profiler {
    int phase = -1;
    int max_samples = 0;
    int count = 0;
    int calls = 0;  // launches of the profiled phase seen since gf_profile_begin (every g_options[GF_OPT_PROFILE_STRIDE]-th is stamped)
    std::vector<hipEvent_t> events;
};
extern Profiler g_prof;
extern int g_options[GF_OPT_COUNT];

// RAII scope of one profiled launch.  Two modes:
//  * bracket (default): an event pair is recorded on `stream` around the launch (≈ 3 µs of event overhead inside the pair);
//  * dispatch (use_dispatch_events()): the launch site passes start()/stop() to hipExtLaunchKernelGGL, which stamps them
//    with the dispatch's own begin/end timestamps — the same clock rocprofv3's kernel trace reports, so the two agree.
struct PhaseScope {
    int idx = -1;
    bool dispatch = false;
    hipStream_t stream;
    PhaseScope(int phase, hipStream_t s) : stream(s) {
        Profiler& p = g_prof;
        if (p.phase == phase) {
            const int stride = g_options[GF_OPT_PROFILE_STRIDE] > 1 ? g_options[GF_OPT_PROFILE_STRIDE] : 1;
            if ((p.calls++ % stride) == 0 && p.count < p.max_samples) idx = p.count;
        }
    }
    bool active() const { return idx >= 0; }
    void use_dispatch_events() { dispatch = true; }
    hipEvent_t start() const { return g_prof.events[2 * idx]; }
    hipEvent_t stop() const { return g_prof.events[2 * idx + 1]; }
    void begin_bracket() {
        if (idx >= 0 && !dispatch) hipEventRecord(g_prof.events[2 * idx], stream);
    }
    ~PhaseScope() {
        if (idx >= 0) {
            Profiler& p = g_prof;
            if (!dispatch) hipEventRecord(p.events[2 * idx + 1], stream);
            p.count = idx + 1;
        }
    }
};

// Launch `kernel` either plainly or, when `scope` is profiling this phase, with dispatch-timestamp events.
#define GF_LAUNCH(scope, kernel, grid, block, lds, strm, ...)                                               \
    do {                                                                                                    \
        if ((scope).active()) {                                                                             \
            (scope).use_dispatch_events();                                                                  \
            hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, strm, (scope).start(), (scope).stop(), 0, __VA_ARGS__); \
        } else {                                                                                            \
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, strm, __VA_ARGS__);                    \
        }                                                                                                   \
    } while (0)

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GF_OK : (int)e;
}

}  // namespace gf
