// launch_cost.hip — host-side cost of enqueueing kernels on this box (development tool).
//   hipcc --offload-arch=gfx950 -O2 -o tools/launch_cost tools/launch_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char b[3800]; };
struct Small { void* p[8]; };
__global__ void k_small(Small s) {}
__global__ void k_big(Big s) {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
template <class F> double burst(F&& f, int n = 60, int reps = 40) {
    double best = 1e9;
    for (int r = 0; r < reps; ++r) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) f();
        auto t1 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
        if (us < best) best = us;
    }
    return best;
}
int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Small sm{}; Big bg{};
    printf("small-arg launch <<<>>>        %.2f us\n", burst([&] { k_small<<<64, 64, 0, s>>>(sm); }));
    printf("3.8KB-arg launch <<<>>>        %.2f us\n", burst([&] { k_big<<<64, 64, 0, s>>>(bg); }));
    printf("1024-WG small launch           %.2f us\n", burst([&] { k_small<<<1024, 256, 0, s>>>(sm); }));
    {   // pre-resolved function handle + hipModuleLaunchKernel (skips the host-stub -> device-function lookup of hipLaunchKernel)
        hipFunction_t fs = nullptr, fb = nullptr;
        CK(hipGetFuncBySymbol(&fs, reinterpret_cast<const void*>(k_small)));
        CK(hipGetFuncBySymbol(&fb, reinterpret_cast<const void*>(k_big)));
        void* ps[] = {&sm};
        void* pb[] = {&bg};
        printf("small-arg hipModuleLaunchKernel %.2f us\n", burst([&] { hipModuleLaunchKernel(fs, 64, 1, 1, 64, 1, 1, 0, s, ps, nullptr); }));
        printf("3.8KB-arg hipModuleLaunchKernel %.2f us\n", burst([&] { hipModuleLaunchKernel(fb, 64, 1, 1, 64, 1, 1, 0, s, pb, nullptr); }));
        printf("small-arg hipLaunchKernel       %.2f us\n", burst([&] { hipLaunchKernel(reinterpret_cast<const void*>(k_small), dim3(64), dim3(64), ps, 0, s); }));
    }
    // graph of 3 kernels
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    k_small<<<64, 64, 0, s>>>(sm); k_big<<<64, 64, 0, s>>>(bg); k_small<<<64, 64, 0, s>>>(sm);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf("graph launch (3 kernel nodes)  %.2f us\n", burst([&] { hipGraphLaunch(ge, s); }));
    // graph + per-launch kernel node param update on all three nodes
    hipGraphNode_t nodes[8]; size_t nn = 8;
    CK(hipGraphGetNodes(g, nodes, &nn));
    hipKernelNodeParams kp[3];
    for (size_t i = 0; i < nn && i < 3; ++i) CK(hipGraphKernelNodeGetParams(nodes[i], &kp[i]));
    printf("graph launch + 3 SetParams     %.2f us\n", burst([&] {
        for (size_t i = 0; i < nn && i < 3; ++i) hipGraphExecKernelNodeSetParams(ge, nodes[i], &kp[i]);
        hipGraphLaunch(ge, s);
    }));
    return 0;
}
