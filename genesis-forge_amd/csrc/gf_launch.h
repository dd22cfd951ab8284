// gf_launch.h — host-side launch helpers: grid sizing and the opt-in per-phase event profiler.
#pragma once

#include <hip/hip_ext.h>

#include <tuple>
#include <utility>
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

// ---------------------------------------------------------------------------------------------
// Launch sink: every kernel launch of the library goes through klaunch().  Normally it is a plain hipLaunchKernel; while
// gf_run_ops_graph is BUILDING a recorded step's hipGraph it appends a kernel node instead, and while it is UPDATING an
// instantiated graph it refreshes that node's arguments (hipGraphExecKernelNodeSetParams) — so the entry points keep
// doing their validation / packing per step (descriptors change: action pointer, RNG streams, ring slots) and only the
// enqueue itself is replaced: one hipGraphLaunch per step instead of one launch per kernel.
// ---------------------------------------------------------------------------------------------
struct NodeShape {
    const void* func;
    dim3 grid, block;
    unsigned lds;
    bool operator==(const NodeShape& o) const {
        return func == o.func && grid.x == o.grid.x && grid.y == o.grid.y && grid.z == o.grid.z && block.x == o.block.x &&
               block.y == o.block.y && block.z == o.block.z && lds == o.lds;
    }
};
struct GraphCache {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<hipGraphNode_t> nodes;
    std::vector<NodeShape> shapes;
};
enum SinkMode { SINK_DIRECT = 0, SINK_BUILD = 1, SINK_UPDATE = 2 };
struct LaunchSink {
    int mode = SINK_DIRECT;
    GraphCache* g = nullptr;
    size_t cursor = 0;
    bool mismatch = false;   // UPDATE: this step's launch sequence differs from the recorded graph
    hipError_t error = hipSuccess;
};
extern thread_local LaunchSink g_sink;

inline void sink_launch(const void* func, dim3 grid, dim3 block, size_t lds, hipStream_t s, void** args) {
    LaunchSink& k = g_sink;
    if (k.mode == SINK_DIRECT) {
        (void)hipLaunchKernel(func, grid, block, args, lds, s);
        return;
    }
    hipKernelNodeParams np{};
    np.func = const_cast<void*>(func);
    np.gridDim = grid;
    np.blockDim = block;
    np.sharedMemBytes = (unsigned)lds;
    np.kernelParams = args;
    np.extra = nullptr;
    const NodeShape shape{func, grid, block, (unsigned)lds};
    if (k.mode == SINK_BUILD) {
        hipGraphNode_t node = nullptr;
        const hipGraphNode_t* dep = k.g->nodes.empty() ? nullptr : &k.g->nodes.back();  // one stream: a linear chain
        const hipError_t e = hipGraphAddKernelNode(&node, k.g->graph, dep, dep ? 1 : 0, &np);
        if (e != hipSuccess) { k.error = e; return; }
        k.g->nodes.push_back(node);
        k.g->shapes.push_back(shape);
        return;
    }
    if (k.cursor >= k.g->nodes.size() || !(k.g->shapes[k.cursor] == shape)) { k.mismatch = true; ++k.cursor; return; }
    const hipError_t e = hipGraphExecKernelNodeSetParams(k.g->exec, k.g->nodes[k.cursor], &np);
    if (e != hipSuccess) k.error = e;
    ++k.cursor;
}

// kernel<<<grid, block, lds, stream>>>(args...) through the sink.  The arguments are converted to the kernel's parameter
// types first (hipLaunchKernel / graph nodes take an array of pointers to exactly those types).
template <class... P, class... A>
inline void klaunch(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t s, A&&... a) {
    static_assert(sizeof...(P) == sizeof...(A), "argument count");
    std::tuple<P...> vals(std::forward<A>(a)...);
    void* ptrs[sizeof...(P) ? sizeof...(P) : 1];
    std::apply([&](auto&... v) { size_t i = 0; ((ptrs[i++] = (void*)&v), ...); }, vals);
    sink_launch(reinterpret_cast<const void*>(kernel), grid, block, lds, s, ptrs);
}

// Launch `kernel` either plainly or, when `scope` is profiling this phase, with dispatch-timestamp events.
#define GF_LAUNCH(scope, kernel, grid, block, lds, strm, ...)                                               \
    do {                                                                                                    \
        if ((scope).active()) {                                                                             \
            (scope).use_dispatch_events();                                                                  \
            hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, strm, (scope).start(), (scope).stop(), 0, __VA_ARGS__); \
        } else {                                                                                            \
            gf::klaunch(kernel, dim3(grid), dim3(block), lds, strm, __VA_ARGS__);                           \
        }                                                                                                   \
    } while (0)

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GF_OK : (int)e;
}

// gait states whose swing / stance bytes chain A left for a later masked gait launch of the same gf_run_ops call (gf_chain.hip)
struct DeferredFlags {
    const float* state[4] = {nullptr, nullptr, nullptr, nullptr};
    bool push(const float* p) {
        for (auto& q : state)
            if (!q || q == p) { q = p; return true; }
        return false;
    }
    bool has(const float* p) const {
        for (auto q : state)
            if (q == p && p) return true;
        return false;
    }
    void pop(const float* p) {
        for (auto& q : state)
            if (q == p) q = nullptr;
    }
    bool empty() const { return !state[0] && !state[1] && !state[2] && !state[3]; }
};

}  // namespace gf
