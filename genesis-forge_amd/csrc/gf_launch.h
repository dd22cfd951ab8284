// gf_launch.h — host-side launch helpers: grid sizing and the opt-in per-phase event profiler.
#pragma once

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
    std::vector<hipEvent_t> events;
};
extern Profiler g_prof;

// RAII bracket: records an event pair on `stream` around the launch when profiling `phase`.
struct PhaseScope {
    int idx = -1;
    hipStream_t stream;
    PhaseScope(int phase, hipStream_t s) : stream(s) {
        Profiler& p = g_prof;
        if (p.phase == phase && p.count < p.max_samples) {
            idx = p.count;
            hipEventRecord(p.events[2 * idx], stream);
        }
    }
    ~PhaseScope() {
        if (idx >= 0) {
            Profiler& p = g_prof;
            hipEventRecord(p.events[2 * idx + 1], stream);
            p.count = idx + 1;
        }
    }
};

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GF_OK : (int)e;
}

}  // namespace gf
