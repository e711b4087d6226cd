// tools/ubench/streamk_proxy.cpp -- what could a stream-K split of the bf16 fc2 (12032 x 1024 x 4096) reach?  (round 6; round-5 verdict "what's missing" 3)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I parakeet.cpp_amd/csrc tools/ubench/streamk_proxy.cpp -o tools/ubench/streamk_proxy
// fc2 today: 252 tiles of 192 x 256, one per CU, 64 K steps each.  Stream-K on 256 x 256 tiles: 188 tiles x 64 = 12 032 K steps over 256 CUs = 47 each.  The proxy
// runs the SAME kernel family on 256 tiles of 256 x 256 with K = 47 x 64 = 3008: every CU does exactly the 47 K steps of the stream-K walk on the bigger tile -- its
// time is the walk's main loop + one epilogue, WITHOUT the partial-tile exchange (one 256 KB fp32 partial per CU written and read once: 2 x 65 MB more).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels/gemm.hip"
#include "kernels/gemm_smallm.hip"
#include "kernels/gemm_smallm_bf16.hip"
using namespace pk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main() {
    const int MM = 16384, NN = 1024, KK = 4096;
    std::vector<unsigned short> h((size_t)MM * KK);
    unsigned s = 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; const float f = ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f; unsigned u; std::memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    void *dA, *dW; float *dB, *dO, *dR;
    CK(hipMalloc(&dA, (size_t)MM * KK * 2)); CK(hipMalloc(&dW, (size_t)NN * KK * 2)); CK(hipMalloc(&dB, NN * 4)); CK(hipMalloc(&dO, (size_t)MM * NN * 4)); CK(hipMalloc(&dR, (size_t)MM * NN * 4));
    CK(hipMemcpy(dA, h.data(), (size_t)MM * KK * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, h.data(), (size_t)NN * KK * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dB, 0, NN * 4)); CK(hipMemset(dR, 0, (size_t)MM * NN * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, int M, int N, int K, int epi) {
        GemmArgs g{reinterpret_cast<const float *>(dA), K, reinterpret_cast<const float *>(dW), K, dB, dO, N, epi == EPI_RESID ? dR : nullptr, N, 0.5f, M, N, K};
        g.a_bf16 = 1; g.fast_act = 1;
        for (int i = 0; i < 5; ++i) launch_gemm_bf16(g, epi, 0);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 30; ++i) launch_gemm_bf16(g, epi, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-78s %7.1f us  %6.0f TF\n", name, ms * 1e3 / 30, 2.0 * M * N * K / (ms / 30) * 1e-9);
    };
    for (int rep = 0; rep < 2; ++rep) {
        run("fc2 as shipped: 12032 x 1024 x 4096, 252 tiles of 192 x 256, residual (register epilogue)", 12032, 1024, 4096, EPI_RESID);
        run("the same product without the residual (fp32 rows out)", 12032, 1024, 4096, EPI_NONE);
        run("stream-K walk proxy: 256 tiles of 256 x 256, K = 3008 (47 K steps per CU), fp32 rows out", 16384, 1024, 3008, EPI_NONE);
        run("  ... with the residual epilogue", 16384, 1024, 3008, EPI_RESID);
        run("out_proj as shipped: 12032 x 1024 x 1024, residual", 12032, 1024, 1024, EPI_RESID);
        run("stream-K walk proxy for out_proj: 256 tiles of 256 x 256, K = 768 (12 K steps per CU)", 16384, 1024, 768, EPI_RESID);
    }
    return 0;
}
