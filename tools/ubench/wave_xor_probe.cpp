// tools/ubench/wave_xor_probe.cpp -- pins the lane exchange pk_devmath.h's wave_xor<OFF>() builds from DPP modifiers and the gfx950 permlane swaps against
// __shfl_xor (ds_bpermute_b32) for OFF = 1, 2, 4, 8, 16, 32 on random words, then times a dependent 6-step butterfly (the canonical sum64 / max64 tree) in both
// forms.   hipcc --offload-arch=gfx950 -O3 -I parakeet.cpp_amd/csrc tools/ubench/wave_xor_probe.cpp -o tools/ubench/wave_xor_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "pk_devmath.h"
using namespace pk;

__global__ void probe(const float *in, float *out) {
    const int lane = threadIdx.x;
    const float v = in[lane];
    out[0 * 64 + lane] = wave_xor<1>(v);
    out[1 * 64 + lane] = wave_xor<2>(v);
    out[2 * 64 + lane] = wave_xor<4>(v);
    out[3 * 64 + lane] = wave_xor<8>(v);
    out[4 * 64 + lane] = wave_xor<16>(v);
    out[5 * 64 + lane] = wave_xor<32>(v);
    out[6 * 64 + lane] = wave_sum64(v);
    out[7 * 64 + lane] = wave_max64(v);
    float p = v;
    for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
    out[8 * 64 + lane] = p;
}
template <bool DPP> __global__ void timing(const float *in, float *out, long long *clk, int n) {
    float p = in[threadIdx.x];
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if constexpr (DPP) p = wave_sum64(p * 0.015625f);
        else { p = p * 0.015625f; for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64); }
    }
    const long long t1 = clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) clk[DPP] = t1 - t0;
}
int main() {
    float h[64], *d_in, *d_out, o[9 * 64];
    long long *d_clk, clk[2];
    srand(7);
    for (int i = 0; i < 64; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    if (hipMalloc(&d_in, sizeof h) != hipSuccess) { printf("no device\n"); return 2; }
    hipMalloc(&d_out, sizeof o); hipMalloc(&d_clk, sizeof clk);
    hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out);
    hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    const int offs[6] = {1, 2, 4, 8, 16, 32};
    for (int k = 0; k < 6; ++k) {
        int b = 0;
        for (int l = 0; l < 64; ++l) b += o[k * 64 + l] != h[l ^ offs[k]];
        printf("wave_xor<%2d>: %s\n", offs[k], b ? "MISMATCH" : "ok");
        bad += b;
    }
    int bs = 0;
    for (int l = 0; l < 64; ++l) bs += o[6 * 64 + l] != o[8 * 64 + l];
    printf("wave_sum64 vs the __shfl_xor tree, bit for bit: %s\n", bs ? "MISMATCH" : "ok");
    bad += bs;
    const int n = 1000;
    hipLaunchKernelGGL(timing<false>, dim3(1), dim3(64), 0, 0, d_in, d_out, d_clk, n);
    hipLaunchKernelGGL(timing<true>, dim3(1), dim3(64), 0, 0, d_in, d_out, d_clk, n);
    hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost);
    printf("dependent 6-step butterfly: __shfl_xor %.1f clocks, wave_xor %.1f clocks per tree\n", (double)clk[0] / n, (double)clk[1] / n);
    return bad ? 1 : 0;
}
