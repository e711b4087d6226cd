// parakeet.cpp_amd/csrc/kernels/subsample.hip -- the depthwise-separable conv subsampling stack
// (reference ConvSubsampling::forward, src/encoder.cpp:219-241).  Activations are channels-last
// ([B][H][W][C]) so the 1x1 convolutions are plain row-major GEMMs on the MFMA kernel and the
// 3x3 depthwise taps are coalesced along C.
//
//  sub_conv1_dw1:  Conv2d(1->C,3x3,s2,p1)+ReLU fused with the following depthwise 3x3 s2: the
//                  [B][C][501][40] conv1 output (1.3 GB at B=64) is never written to HBM; each
//                  depthwise output recomputes its 3x3 neighbourhood of conv1 values from the 7x7
//                  input patch (81 fma instead of a 2.6 GB round trip).
//  sub_dw:         depthwise 3x3 s2 p1 on a channels-last tensor.
// Tap order (ky,kx) and "bias after the chain" follow the oracle exactly (bit-identical results).
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

__global__ __launch_bounds__(256) void sub_conv1_dw1_kernel(const float *__restrict__ feats, int Tm, int F, int C,
                                                            const float *__restrict__ w1 /*[9][C]*/, const float *__restrict__ b1,
                                                            const float *__restrict__ wd /*[9][C]*/, const float *__restrict__ bd,
                                                            int H1, int W1, int H2, int W2, int64_t n_pix, float *__restrict__ out) {
    const int ppb = 256 / C;                                   // pixels per block (C <= 256, C | 256)
    const int c = threadIdx.x % C;
    const int64_t pix = (int64_t)blockIdx.x * ppb + threadIdx.x / C;
    if (pix >= n_pix) return;
    const int x2 = (int)(pix % W2);
    const int y2 = (int)((pix / W2) % H2);
    const int b = (int)(pix / ((int64_t)W2 * H2));
    const float *in = feats + (int64_t)b * Tm * F;
    float k1[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k1[i] = w1[i * C + c];
    const float bias1 = b1[c];
    float acc2 = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int y1 = 2 * y2 + ky - 1;
        if (y1 < 0 || y1 >= H1) continue;                      // zero padding of the depthwise conv
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int x1 = 2 * x2 + kx - 1;
            if (x1 < 0 || x1 >= W1) continue;
            float acc1 = 0.0f;                                 // conv1 at (y1, x1), src/encoder.cpp:223
#pragma unroll
            for (int jy = 0; jy < 3; ++jy) {
                const int iy = 2 * y1 + jy - 1;
                if (iy < 0 || iy >= Tm) continue;
#pragma unroll
                for (int jx = 0; jx < 3; ++jx) {
                    const int ix = 2 * x1 + jx - 1;
                    if (ix < 0 || ix >= F) continue;
                    acc1 = __builtin_fmaf(k1[jy * 3 + jx], in[(int64_t)iy * F + ix], acc1);
                }
            }
            float v = acc1 + bias1;
            v = v > 0.0f ? v : 0.0f;                           // ReLU :224
            acc2 = __builtin_fmaf(wd[(ky * 3 + kx) * C + c], v, acc2);   // dw1 :226
        }
    }
    out[pix * C + c] = acc2 + bd[c];
}

__global__ __launch_bounds__(256) void sub_dw_kernel(const float *__restrict__ in, int H, int W, int C,
                                                     const float *__restrict__ wd /*[9][C]*/, const float *__restrict__ bd,
                                                     int Ho, int Wo, int64_t n_pix, float *__restrict__ out) {
    const int ppb = 256 / C;
    const int c = threadIdx.x % C;
    const int64_t pix = (int64_t)blockIdx.x * ppb + threadIdx.x / C;
    if (pix >= n_pix) return;
    const int xo = (int)(pix % Wo);
    const int yo = (int)((pix / Wo) % Ho);
    const int b = (int)(pix / ((int64_t)Wo * Ho));
    const float *src = in + (int64_t)b * H * W * C;
    float acc = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * yo + ky - 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * xo + kx - 1;
            if (ix < 0 || ix >= W) continue;
            acc = __builtin_fmaf(wd[(ky * 3 + kx) * C + c], src[((int64_t)iy * W + ix) * C + c], acc);
        }
    }
    out[pix * C + c] = acc + bd[c];
}

void launch_sub_conv1_dw1(const float *feats, int B, int Tm, int F, int C, const float *w1, const float *b1, const float *wd,
                          const float *bd, float *out, hipStream_t s) {
    const int H1 = (Tm - 1) / 2 + 1, W1 = (F - 1) / 2 + 1, H2 = (H1 - 1) / 2 + 1, W2 = (W1 - 1) / 2 + 1;
    const int64_t n_pix = (int64_t)B * H2 * W2;
    const int ppb = 256 / C;
    hipLaunchKernelGGL(sub_conv1_dw1_kernel, dim3((unsigned)((n_pix + ppb - 1) / ppb)), dim3(256), 0, s, feats, Tm, F, C, w1, b1,
                       wd, bd, H1, W1, H2, W2, n_pix, out);
}
void launch_sub_dw(const float *in, int B, int H, int W, int C, const float *wd, const float *bd, float *out, hipStream_t s) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t n_pix = (int64_t)B * Ho * Wo;
    const int ppb = 256 / C;
    hipLaunchKernelGGL(sub_dw_kernel, dim3((unsigned)((n_pix + ppb - 1) / ppb)), dim3(256), 0, s, in, H, W, C, wd, bd, Ho, Wo,
                       n_pix, out);
}

}  // namespace pk
