// tools/ubench/smallm_chain.cpp -- where does the small-M GEMM (kernels/gemm_smallm.hip) spend its time?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I parakeet.cpp_amd/csrc tools/ubench/smallm_chain.cpp -o /tmp/smallm_chain && /tmp/smallm_chain
// (1) a bare dependent chain of v_mfma_f32_16x16x4_f32 (no memory): clocks per MFMA and the shader clock while only ~128 waves run;
// (2) the shipped kernel on the streaming shapes (M = 32) with COLD weights (24 different matrices in turn, as the encoder layers) and
//     with a HOT one (the same matrix every launch);
// (3) the same with W rows padded by 256 bytes (row pitch no longer a multiple of 16 KB: does the HBM channel mapping matter?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels/gemm_smallm.hip"

using namespace pk;
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void chain_kernel(int n, float *out, long long *clk) {
    f4 acc = {0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    const long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    const long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = acc[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

static float time_launches(int reps, const std::vector<GemmArgs> &gs, int epi) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t i = 0; i < gs.size(); ++i) launch_gemm_smallm(gs[i], epi, 0);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch_gemm_smallm(gs[r % gs.size()], epi, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0f / reps;
}

// the chain with the operand transposes of the shipped kernel in front of every four MFMAs (operands from registers, no memory)
__global__ void chain_swap_kernel(int n, float *out, long long *clk) {
    f4 acc = {0, 0, 0, 0};
    float4 a0 = {threadIdx.x * 1e-3f, 1.0f, 2.0f, 3.0f}, w0 = {1.0f, 0.5f, 0.25f, 0.125f};
    const long long t0 = __builtin_readcyclecounter(), w0c = wall_clock64();
    for (int i = 0; i < n; i += 4) {
        float4 a = a0, w = w0;
        a.x += (float)i;                          // new operand values every block, as fresh loads would be
        sm_tr4x4(a);
        sm_tr4x4(w);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w.w, acc, 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = acc[0];
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0c; }
}

int main() {
    float *out; long long *clk, h[2];
    hipMalloc(&out, 1 << 20); hipMalloc(&clk, 16);
    for (int waves : {1, 128, 1024}) {
        hipLaunchKernelGGL(chain_kernel, dim3(waves), dim3(64), 0, 0, 4096, out, clk);
        hipLaunchKernelGGL(chain_kernel, dim3(waves), dim3(64), 0, 0, 4096, out, clk);
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        printf("chain of 4096 MFMA 16x16x4 f32, %4d waves: %.1f shader clocks per MFMA, %.2f ns per MFMA (wall clock at 100 MHz)\n", waves, h[0] / 4096.0,
               h[1] * 10.0 / 4096.0);
    }
    hipLaunchKernelGGL(chain_swap_kernel, dim3(128), dim3(64), 0, 0, 4096, out, clk);
    hipLaunchKernelGGL(chain_swap_kernel, dim3(128), dim3(64), 0, 0, 4096, out, clk);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("the same chain with 8 permlane swaps per 4 MFMAs, 128 waves: %.1f shader clocks per MFMA, %.2f ns per MFMA\n", h[0] / 4096.0, h[1] * 10.0 / 4096.0);
    const int M = 32, NL = 24;
    struct Shape { const char *name; int N, K, epi; } shapes[] = {
        {"fc2   N 1024 K 4096 resid", 1024, 4096, EPI_RESID}, {"fc1   N 4096 K 1024 silu ", 4096, 1024, EPI_SILU},
        {"proj  N 1024 K 1024 resid", 1024, 1024, EPI_RESID}, {"qkv   N 3072 K 1024 none ", 3072, 1024, EPI_NONE},
        {"glu   N 1024 K 1024 glu  ", 1024, 1024, EPI_GLU}};
    for (auto &sh : shapes) {
        for (int pad : {0, 64}) {
            const int rows = sh.epi == EPI_GLU ? 2 * sh.N : sh.N, ldw = sh.K + pad;
            float *A, *W, *O, *bias;
            hipMalloc(&A, (size_t)M * sh.K * 4); hipMalloc(&W, (size_t)NL * rows * ldw * 4); hipMalloc(&O, (size_t)M * sh.N * 4); hipMalloc(&bias, rows * 4);
            hipMemset(A, 0, (size_t)M * sh.K * 4); hipMemset(W, 0, (size_t)NL * rows * ldw * 4); hipMemset(O, 0, (size_t)M * sh.N * 4); hipMemset(bias, 0, rows * 4);
            std::vector<GemmArgs> cold, hot;
            for (int l = 0; l < NL; ++l) {
                GemmArgs g{A, sh.K, W + (size_t)l * rows * ldw, ldw, bias, O, sh.N, O, sh.N, 1.0f, M, sh.N, sh.K};
                cold.push_back(g);
                if (l == 0) hot.push_back(g);
            }
            const float tc = time_launches(480, cold, sh.epi), th = time_launches(480, hot, sh.epi);
            for (auto &g : cold) { g.a_sigma = 1; g.W_sig = g.W; }     // timing only: the operands are zeros, the layout does not matter
            const float ts = time_launches(480, cold, sh.epi);
            const double mb = (double)rows * sh.K * 4 / 1e6;
            printf("%s  W pitch %5d B: cold %6.2f us (%.2f TB/s)  hot %6.2f us  sigma operands (cold) %6.2f us  [bare chain at 44 clk / 2.4 GHz: %.1f us]\n", sh.name, ldw * 4, tc, mb / tc,
                   th, ts, sh.K / 4 * 44 / 2400.0);
            hipFree(A); hipFree(W); hipFree(O); hipFree(bias);
        }
    }
    return 0;
}
