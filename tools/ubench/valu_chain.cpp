// tools/ubench/valu_chain.cpp -- latency of a DEPENDENT fp32 fma chain on gfx950: v_fma_f32 (one chain per lane) against v_mfma_f32_16x16x4_f32
// (one chain per 16x16 tile, 4 k per instruction).  The small-M products of the streaming encoder are latency-bound chains of K dependent
// steps (natural-k contract); this measures the floor per k for both instruction kinds, one wave per SIMD and several.
// usage: valu_chain [K=4096]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void valu_kernel(const float *in, float *out, int K, long long *clk) {
    float a[CHAINS], acc[CHAINS];
    const float w = in[threadIdx.x & 63];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { a[c] = in[64 + c + threadIdx.x]; acc[c] = 0.0f; }
    __syncthreads();
    const long long t0 = clock64();
    for (int k = 0; k < K; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_fmaf(a[c], w, acc[c]);
        asm volatile("" ::: "memory");
    }
    const long long t1 = clock64();
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void mfma_kernel(const float *in, float *out, int K, long long *clk) {
    const float a = in[threadIdx.x & 63], w = in[64 + (threadIdx.x & 63)];
    f32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = clock64();
    for (int k = 0; k < K; k += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w, acc, 0, 0, 0);
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 4096;
    float *in, *out; long long *clk;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&clk, 8));
    CK(hipMemset(in, 0, 4096 * 4));
    auto run = [&](const char *name, auto launch, int kper) {
        for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
            const int threads = 64 * 4 * waves_per_simd;          // one workgroup per CU, 4 SIMDs
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            launch(threads); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); launch(threads); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
            printf("  %-44s %d wave(s)/SIMD: %7.2f shader clocks per k-step (%lld clocks for K = %d; kernel %.1f us)\n", name, waves_per_simd, (double)c / K, c, K, ms * 1e3);
        }
        (void)kper;
    };
    printf("dependent fma chains, K = %d (clock64 = shader clock)\n", K);
    run("v_fma_f32, 1 chain per lane", [&](int t) { hipLaunchKernelGGL(valu_kernel<1>, dim3(256), dim3(t), 0, 0, in, out, K, clk); }, 1);
    run("v_fma_f32, 2 interleaved chains per lane", [&](int t) { hipLaunchKernelGGL(valu_kernel<2>, dim3(256), dim3(t), 0, 0, in, out, K, clk); }, 1);
    run("v_fma_f32, 4 interleaved chains per lane", [&](int t) { hipLaunchKernelGGL(valu_kernel<4>, dim3(256), dim3(t), 0, 0, in, out, K, clk); }, 1);
    run("v_mfma_f32_16x16x4_f32, 1 chain per wave", [&](int t) { hipLaunchKernelGGL(mfma_kernel, dim3(256), dim3(t), 0, 0, in, out, K, clk); }, 4);
    return 0;
}
