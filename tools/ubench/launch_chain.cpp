// tools/ubench/launch_chain.cpp -- what ONE dependent launch costs on an idle MI355X as a function of the kernel's own duration: a chain of N launches on one
// stream, each spinning for X ns (wall clock) after reading the word the previous launch wrote; plain launches and a hipGraph of 64 nodes; 1 / 160 workgroups;
// small / 400-byte kernel arguments (the decode loop's records are ~300-400 bytes).   hipcc --offload-arch=gfx950 -O3 -o launch_chain launch_chain.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Big { int *w; int ns; int pad[96]; };
__global__ void spin_small(int *w, int ns) {
    const int v = w[blockIdx.x];
    const long long t0 = wall_clock64();
    while ((wall_clock64() - t0) * 10 < ns) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 0) w[blockIdx.x] = v + 1;
}
__global__ void spin_big(Big a) {
    const int v = a.w[blockIdx.x];
    const long long t0 = wall_clock64();
    while ((wall_clock64() - t0) * 10 < a.ns) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 0) a.w[blockIdx.x] = v + 1 + a.pad[5];
}
int main() {
    int *w; CK(hipMalloc(&w, 4096 * 4)); CK(hipMemset(w, 0, 4096 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int N = 1024;
    printf("per-launch us on a chain of %d dependent launches (kernel spins X ns)\n", N);
    printf("%8s %6s %6s | %9s %9s\n", "X_ns", "grid", "args", "launches", "graph64");
    for (int big = 0; big < 2; ++big)
    for (int grid : {1, 160})
    for (int ns : {0, 1000, 2000, 3000, 4000, 6000, 8000}) {
        Big a{}; a.w = w; a.ns = ns;
        auto one = [&]() { if (big) hipLaunchKernelGGL(spin_big, dim3(grid), dim3(256), 0, s, a); else hipLaunchKernelGGL(spin_small, dim3(grid), dim3(256), 0, s, w, ns); };
        for (int i = 0; i < 64; ++i) one();
        CK(hipStreamSynchronize(s));
        double best_l = 1e9, best_g = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) one();
            CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (us < best_l) best_l = us;
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 64; ++i) one();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N / 64; ++i) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (us < best_g) best_g = us;
        }
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        printf("%8d %6d %6s | %9.2f %9.2f\n", ns, grid, big ? "400B" : "16B", best_l, best_g);
    }
    return 0;
}
