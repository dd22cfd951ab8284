// microbench_observe.hip — times gf_observe alone on history observations (development tool).
//   policy-like: 14 strided gait columns + the 48-wide Go2 frame, O = 62, H = 5      critic-like: 4 contact norms + 12 dof forces, O = 16, H = 5
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_observe tools/microbench_observe.hip -Lgenesis-forge_amd -lgf_step -Wl,-rpath,'$ORIGIN/../genesis-forge_amd'
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/gf_step.h"

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

static float* dallocf(size_t n, float fill = 0.f) {
    float* p;
    CK(hipMalloc(&p, n * sizeof(float)));
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = fill + (float)((i * 2654435761u) & 1023) * 1e-3f;
    CK(hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    return p;
}

__global__ __launch_bounds__(256) void shifted_copy(const float* __restrict__ in, float* __restrict__ out, size_t n4, int shift) {
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f4a __attribute__((ext_vector_type(4)));
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4 && i * 4 >= (size_t)shift) {
        const f4u v = *reinterpret_cast<const f4u*>(in + i * 4 - shift);
        *reinterpret_cast<f4a*>(out + i * 4) = f4a{v.x, v.y, v.z, v.w};
    }
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
    printf("%-34s %8.2f us/launch  %8.1f GB/s (algorithmic)\n", name, us, bytes / us / 1e3);
    return us;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 65536;
    const int iters = argc > 2 ? atoi(argv[2]) : 300;
    const int D = 12, H = 5, L = 4;
    printf("N=%d iters=%d\n", N, iters);
    float *pos = dallocf((size_t)N * 3), *quat = dallocf((size_t)N * 4, 0.5f), *lin = dallocf((size_t)N * 3), *ang = dallocf((size_t)N * 3);
    float *dof = dallocf((size_t)N * D), *dvel = dallocf((size_t)N * D), *targets = dallocf((size_t)N * D), *force = dallocf((size_t)N * D);
    float *cmd = dallocf((size_t)N * 3), *gait = dallocf((size_t)N * GF_GAIT_ROW), *contacts = dallocf((size_t)N * L * 3);

    GfObservationArgs pol{};
    pol.num_envs = N; pol.num_dofs = D; pol.history_len = H; pol.entity = {pos, quat, lin, ang};
    pol.dof_pos = dof; pol.dof_vel = dvel; pol.targets = targets;
    pol.command[0].command = gait; pol.command[0].width = GF_GAIT_OBS_WIDTH; pol.command[0].stride = GF_GAIT_ROW;
    pol.command[1].command = cmd; pol.command[1].width = 3;
    pol.seed = 1; pol.stream = 3;
    const int pops[8] = {GF_O_COMMAND, GF_O_COMMAND, GF_O_ANG_VEL_BODY, GF_O_LIN_VEL_BODY, GF_O_PROJ_GRAVITY, GF_O_DOF_POS, GF_O_DOF_VEL, GF_O_ACTIONS};
    const int pw[8] = {14, 3, 3, 3, 3, 12, 12, 12};
    int O = 0;
    for (int k = 0; k < 8; ++k) { pol.items[k].op = pops[k]; pol.items[k].width = pw[k]; pol.items[k].scale = 1.f; O += pw[k]; }
    pol.items[1].i0 = 1; pol.items[6].scale = 0.05f;
    pol.num_items = 8; pol.obs_width = O;
    float* pbuf[2] = {dallocf((size_t)N * O * H), dallocf((size_t)N * O * H)};

    GfObservationArgs cri{};
    cri.num_envs = N; cri.num_dofs = D; cri.history_len = H; cri.dof_force = force;
    cri.contact[0].contacts = contacts; cri.contact[0].num_links = L;
    cri.items[0].op = GF_O_CONTACT_FORCE_NORM; cri.items[0].width = L; cri.items[0].scale = 1.f;
    cri.items[1].op = GF_O_DOF_FORCE; cri.items[1].width = D; cri.items[1].scale = 0.1f;
    cri.num_items = 2; cri.obs_width = L + D; cri.seed = 1; cri.stream = 4;
    float* cbuf[2] = {dallocf((size_t)N * (L + D) * H), dallocf((size_t)N * (L + D) * H)};

    GfObservationArgs one = pol;   // the same frame without history
    one.history_len = 1;

    auto chk = [](int rc, const char* what) { if (rc) { fprintf(stderr, "%s failed: %d %s\n", what, rc, gf_error_string(rc)); exit(2); } };
    const double Nd = N;
    const size_t n4 = (size_t)N * O * H / 4;
    time_loop("shifted copy of [N,310] (by 62)", iters, 8.0 * n4 * 4, [&](int i) { shifted_copy<<<(unsigned)((n4 + 255) / 256), 256>>>(pbuf[i & 1], pbuf[(i + 1) & 1], n4, O); });
    const double pol_bytes = (4.0 * O * (H - 1) + 4.0 * O * H + 4.0 * (14 + 3 + 36 + 10)) * Nd;
    time_loop("gf_observe policy O=62 H=5", iters, pol_bytes, [&](int i) { pol.obs = pbuf[i & 1]; pol.prev_obs = pbuf[(i + 1) & 1]; chk(gf_observe(&pol, 0), "policy"); });
    time_loop("gf_observe policy O=62 H=1", iters, (4.0 * O + 4.0 * 63) * Nd, [&](int i) { one.obs = pbuf[i & 1]; chk(gf_observe(&one, 0), "one"); });
    const double cri_bytes = (4.0 * 16 * (H - 1) + 4.0 * 16 * H + 4.0 * (12 + 12)) * Nd;
    time_loop("gf_observe critic O=16 H=5", iters, cri_bytes, [&](int i) { cri.obs = cbuf[i & 1]; cri.prev_obs = cbuf[(i + 1) & 1]; chk(gf_observe(&cri, 0), "critic"); });
    return 0;
}
