// Micro-benchmark: latency of dependent fp32 MFMA chains and of dependent VALU fma chains (decode-loop sizing).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int N> __global__ void chain16(float *out, float a, float b) {
    f32x4 acc = {0, 0, 0, 0};
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = acc[0]; out[1] = (float)(t1 - t0); }
    else if (acc[0] == 12345.f) out[2] = acc[1];
}
template <int N> __global__ void chain32(float *out, float a, float b) {
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0;
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = acc[0]; out[1] = (float)(t1 - t0); }
    else if (acc[0] == 12345.f) out[2] = acc[1];
}
template <int N> __global__ void chainfma(float *out, float a, float b) {
    float acc = 0;
    long long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) acc = __builtin_fmaf(a, b, acc);
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = acc; out[1] = (float)(t1 - t0); }
    else if (acc == 12345.f) out[2] = acc;
}
template <class F> float timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < reps; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000.f / reps;
}
int main() {
    float *d; hipMalloc(&d, 64); float h[4];
    for (int blocks : {1, 160, 1024}) {
        float us = timeit([&] { hipLaunchKernelGGL(chain16<160>, dim3(blocks), dim3(256), 0, 0, d, 1.f, 2.f); }, 200);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("mfma16x16x4 x160 blocks=%4d : %.2f us/launch, %.0f shader cycles (%.1f cyc/mfma)\n", blocks, us, h[1], h[1] / 160);
        us = timeit([&] { hipLaunchKernelGGL(chain32<320>, dim3(blocks), dim3(256), 0, 0, d, 1.f, 2.f); }, 200);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("mfma32x32x2 x320 blocks=%4d : %.2f us/launch, %.0f shader cycles (%.1f cyc/mfma)\n", blocks, us, h[1], h[1] / 320);
        us = timeit([&] { hipLaunchKernelGGL(chainfma<640>, dim3(blocks), dim3(256), 0, 0, d, 1.f, 2.f); }, 200);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("v_fma        x640 blocks=%4d : %.2f us/launch, %.0f shader cycles (%.1f cyc/fma)\n", blocks, us, h[1], h[1] / 640);
    }
    float us = timeit([&] { hipLaunchKernelGGL(chainfma<1>, dim3(1), dim3(64), 0, 0, d, 1.f, 2.f); }, 500);
    printf("empty-ish kernel: %.2f us/launch (back-to-back launches, same stream)\n", us);
    return 0;
}
