// tools/ubench/gemm_bf16_k.cpp -- fixed cost vs main-loop rate of the bf16 direct-to-LDS GEMM (kernels/gemm_bf16_glds.hpp) on the tdt-600m
// FFN shapes: the same M x N at K = 1024 / 2048 / 4096, with and without the epilogue's activation, fp32 and bf16 outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I parakeet.cpp_amd/csrc tools/ubench/gemm_bf16_k.cpp -o /tmp/gemm_bf16_k && /tmp/gemm_bf16_k
// slope over K = main loop, intercept = prologue + epilogue + tail.  Operands are random bf16 (uniform [-1, 1)): zero-filled operands clock higher.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels/gemm.hip"
#include "kernels/gemm_smallm.hip"

using namespace pk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    const int M = 12032, NMAX = 4096, KMAX = 4096;
    std::vector<unsigned short> h((size_t)M * KMAX);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; const float f = ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f; unsigned u; std::memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (auto &v : h) v = rnd();
    void *dA, *dW; float *dB, *dO, *dR;
    CK(hipMalloc(&dA, (size_t)M * KMAX * 2)); CK(hipMalloc(&dW, (size_t)NMAX * KMAX * 2)); CK(hipMalloc(&dB, NMAX * 4));
    CK(hipMalloc(&dO, (size_t)M * NMAX * 4)); CK(hipMalloc(&dR, (size_t)M * NMAX * 4));
    CK(hipMemcpy(dA, h.data(), (size_t)M * KMAX * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, h.data(), (size_t)NMAX * KMAX * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dB, 0, NMAX * 4)); CK(hipMemset(dR, 0, (size_t)M * NMAX * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char *name; int N, epi, out16, fast; };
    const Case cases[] = {{"N 4096 none  fp32 out", 4096, EPI_NONE, 0, 0}, {"N 4096 none  bf16 out", 4096, EPI_NONE, 1, 0}, {"N 4096 silu  bf16 out (fc1)", 4096, EPI_SILU, 1, 1},
                          {"N 1024 none  fp32 out", 1024, EPI_NONE, 0, 0}, {"N 1024 resid fp32 out (fc2)", 1024, EPI_RESID, 0, 0}};
    for (const Case &c : cases) {
        double t[3];
        int i = 0;
        for (int K : {1024, 2048, 4096}) {
            GemmArgs g{reinterpret_cast<const float *>(dA), K, reinterpret_cast<const float *>(dW), K, dB, dO, c.N, dR, c.N, 0.5f, M, c.N, K};
            g.a_bf16 = 1; g.out_bf16 = c.out16; g.fast_act = c.fast;
            for (int r = 0; r < 3; ++r) launch_gemm_bf16(g, c.epi, 0);
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) launch_gemm_bf16(g, c.epi, 0);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t[i++] = ms * 1e3 / reps;
        }
        const double slope = (t[2] - t[0]) / 3072.0;                 // us per k
        const double tf = 2.0 * M * c.N / slope * 1e-6;              // main-loop TFLOP/s
        printf("%-30s K 1024 %7.1f us (%6.0f TF)  2048 %7.1f  4096 %7.1f us (%6.0f TF) | main loop %6.0f TF, fixed %5.1f us\n", c.name, t[0],
               2.0 * M * c.N * 1024 / t[0] * 1e-6, t[1], t[2], 2.0 * M * c.N * 4096 / t[2] * 1e-6, tf, t[0] - slope * 1024);
    }
    // tile shapes per product: what the dispatcher picks vs the 192 x 256 tile (2 x 4 waves of 96 x 64) whose count lands on one / three rounds of 256 CUs
    struct Prod { const char *name; int N, K, epi, out16; };
    const Prod prods[] = {{"fc2  N 1024 K 4096 resid", 1024, 4096, EPI_RESID, 0}, {"out  N 1024 K 1024 resid", 1024, 1024, EPI_RESID, 0},
                          {"qkv  N 3072 K 1024 none bf16", 3072, 1024, EPI_NONE, 1}, {"fc1  N 4096 K 1024 silu bf16", 4096, 1024, EPI_SILU, 1},
                          {"pw1  N 1024 K 1024 glu", 1024, 1024, EPI_GLU, 0}};
    for (const Prod &c : prods) {
        GemmArgs g{reinterpret_cast<const float *>(dA), c.K, reinterpret_cast<const float *>(dW), c.K, dB, dO, c.N, dR, c.N, 0.5f, M, c.N, c.K};
        g.a_bf16 = 1; g.out_bf16 = c.out16; g.fast_act = c.epi == EPI_SILU || c.epi == EPI_GLU;
        auto timeit = [&](auto &&run) {
            for (int r = 0; r < 3; ++r) run();
            CK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) run();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            return ms * 1e3 / reps;
        };
        const double gf = 2.0 * M * c.N * c.K * 1e-6;
        const double t0 = timeit([&] { launch_gemm_bf16(g, c.epi, 0); });
        double t1 = 0, t2 = 0;
        if (c.epi == EPI_RESID) { t1 = timeit([&] { launch_gemm_bf16_glds<2, 4, 3, 2, EPI_RESID>(g, 0); }); t2 = timeit([&] { launch_gemm_bf16_glds<4, 2, 2, 4, EPI_RESID>(g, 0); }); }
        if (c.epi == EPI_NONE) { t1 = timeit([&] { launch_gemm_bf16_glds<2, 4, 3, 2, EPI_NONE>(g, 0); }); t2 = timeit([&] { launch_gemm_bf16_glds<4, 2, 2, 4, EPI_NONE>(g, 0); }); }
        if (c.epi == EPI_GLU) { t1 = timeit([&] { launch_gemm_bf16_glds<2, 4, 3, 2, EPI_GLU>(g, 0); }); t2 = timeit([&] { launch_gemm_bf16_glds<4, 2, 2, 4, EPI_GLU>(g, 0); }); }
        if (c.epi == EPI_SILU) { t1 = timeit([&] { launch_gemm_bf16_glds<2, 4, 3, 2, EPI_SILU>(g, 0); }); t2 = timeit([&] { launch_gemm_bf16_glds<4, 2, 2, 4, EPI_SILU>(g, 0); }); }
        printf("%-30s dispatcher %6.1f us (%5.0f TF)   192x256 %6.1f us (%5.0f TF)   256x256 %6.1f us (%5.0f TF)\n", c.name, t0, gf / t0, t1, gf / t1, t2, gf / t2);
    }
    return 0;
}
