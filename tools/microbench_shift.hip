// microbench_shift.hip — how fast can the pure-history shift of a [N, O*H] observation array go, and what slows it (development tool).
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_shift tools/microbench_shift.hip && tools/microbench_shift [N] [iters]
// Variants: 1 unit per lane (a plain shifted copy) / K units per lane in flight (the HistBatch structure of csrc/gf_obs_hist.h) with
// the pure-unit test, with a dynamic LDS allocation that caps the workgroups per CU, and with K = 2, 4, 8.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void shift1(const float* __restrict__ in, float* __restrict__ out, size_t n4, int O, int OH) {
    extern __shared__ float lds[];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        const size_t e = i * 4;
        const int c = (int)(e % (size_t)OH);
        if (c >= O && c + 3 < OH) {
            const f4u v = *reinterpret_cast<const f4u*>(in + e - O);
            *reinterpret_cast<f4a*>(out + e) = f4a{v.x, v.y, v.z, v.w};
        }
    }
}

template <int K, bool UNCOND>
__global__ __launch_bounds__(256) void shiftK(const float* __restrict__ in, float* __restrict__ out, size_t n4, int O, int OH) {
    extern __shared__ float lds[];
    const size_t u0 = (size_t)blockIdx.x * (K * 256);
    f4u v[K];
    unsigned pure = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t u = u0 + threadIdx.x + k * 256;
        const size_t e = (u < n4 ? u : u0) * 4;
        const int c = (int)(e % (size_t)OH);
        const bool p = u < n4 && c >= O && c + 3 < OH;
        pure |= p ? 1u << k : 0u;
        if (UNCOND) v[k] = *reinterpret_cast<const f4u*>(in + (p ? e - O : u0 * 4));
        else if (p) v[k] = *reinterpret_cast<const f4u*>(in + e - O);
    }
    if (UNCOND) __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int k = 0; k < K; ++k)
        if ((pure >> k) & 1u) *reinterpret_cast<f4a*>(out + (u0 + threadIdx.x + k * 256) * 4) = f4a{v[k].x, v[k].y, v[k].z, v[k].w};
}

// consecutive units per lane (lane owns K*16 contiguous bytes) instead of strided
template <int K>
__global__ __launch_bounds__(256) void shiftK_contig(const float* __restrict__ in, float* __restrict__ out, size_t n4, int O, int OH) {
    const size_t u0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * K;
    f4u v[K];
    unsigned pure = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t u = u0 + k;
        const size_t e = (u < n4 ? u : 0) * 4;
        const int c = (int)(e % (size_t)OH);
        const bool p = u < n4 && c >= O && c + 3 < OH;
        pure |= p ? 1u << k : 0u;
        v[k] = *reinterpret_cast<const f4u*>(in + (p ? e - O : 0));
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int k = 0; k < K; ++k)
        if ((pure >> k) & 1u) *reinterpret_cast<f4a*>(out + (u0 + k) * 4) = f4a{v[k].x, v[k].y, v[k].z, v[k].w};
}

template <typename F>
double time_loop(const char* name, int iters, double bytes, F&& f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) f(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) f(i);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    printf("%-58s %8.2f us/launch  %8.1f GB/s\n", name, us, bytes / us / 1e3);
    return us;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 65536;
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    const int O = 62, H = 5, OH = O * H;
    const size_t n = (size_t)N * OH, n4 = n / 4;
    float* buf[3];
    for (auto& b : buf) { CK(hipMalloc(&b, n * 4 + 4096)); CK(hipMemset(b, 0, n * 4 + 4096)); }
    const double bytes = 8.0 * N * O * (H - 1) * 0.98;   // pure units only (~ the history columns)
    printf("N=%d O=%d H=%d  %.1f MB read + written per launch (three buffers rotate, as the env's output slots do)\n", N, O, H, bytes / 1e6);
    auto in = [&](int i) { return buf[i % 3]; };
    auto out = [&](int i) { return buf[(i + 1) % 3]; };
    time_loop("1 unit/lane", iters, bytes, [&](int i) { shift1<<<(unsigned)((n4 + 255) / 256), 256>>>(in(i), out(i), n4, O, OH); });
    time_loop("1 unit/lane, 32 KB LDS/WG", iters, bytes, [&](int i) { shift1<<<(unsigned)((n4 + 255) / 256), 256, 32768>>>(in(i), out(i), n4, O, OH); });
#define RUNK(K, U, L) time_loop(#K " units/lane strided, " #U ", LDS " #L, iters, bytes, [&](int i) { shiftK<K, U><<<(unsigned)((n4 + K * 256 - 1) / (K * 256)), 256, L>>>(in(i), out(i), n4, O, OH); });
    RUNK(2, true, 0) RUNK(4, true, 0) RUNK(8, true, 0) RUNK(8, false, 0) RUNK(8, true, 32768) RUNK(4, true, 32768) RUNK(2, true, 32768) RUNK(8, true, 40000)
#define RUNC(K) time_loop(#K " units/lane contiguous", iters, bytes, [&](int i) { shiftK_contig<K><<<(unsigned)((n4 + K * 256 - 1) / (K * 256)), 256>>>(in(i), out(i), n4, O, OH); });
    RUNC(2) RUNC(4) RUNC(8)
    return 0;
}
