// parakeet.cpp_amd/csrc/kernels/convmod.hip -- the middle of the Conformer convolution module
// (reference ConformerConvModule::forward, src/encoder.cpp:66-68): depthwise Conv1d(k, pad (k-1)/2,
// groups=d) -> BatchNorm1d (inference, running stats) -> SiLU, fused in one pass over [B][T][d].
// The pointwise convs + GLU on either side are epilogues of the MFMA GEMM.
// One thread per (frame, channel); channels are the fast axis, so the k taps are k coalesced rows.
// BatchNorm is applied as written ((y-mean)*rstd*gamma+beta, rstd precomputed on the host) rather
// than folded into the taps, so the result is bit-identical to the oracle.
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

template <int KC>
__global__ __launch_bounds__(256) void dwconv_bn_silu_kernel(const float *__restrict__ g, int T, int d,
                                                             const float *__restrict__ w /*[KC][d]*/, const float *__restrict__ bias,
                                                             const float *__restrict__ bn_mean, const float *__restrict__ bn_rstd,
                                                             const float *__restrict__ bn_g, const float *__restrict__ bn_b,
                                                             int64_t n, float *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int ch = (int)(idx % d);
    const int64_t row = idx / d;
    const int t = (int)(row % T);
    float acc = 0.0f;
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
        const int tt = t + kk - (KC - 1) / 2;
        if (tt < 0 || tt >= T) continue;                       // zero padding
        acc = __builtin_fmaf(w[kk * d + ch], g[(row + (tt - t)) * d + ch], acc);
    }
    float v = acc + bias[ch];
    v = __builtin_fmaf((v - bn_mean[ch]) * bn_rstd[ch], bn_g[ch], bn_b[ch]);
    out[idx] = dsiluf(v);
}

void launch_dwconv_bn_silu(const float *g, int B, int T, int d, int kc, const float *w, const float *bias, const float *bn_mean,
                           const float *bn_rstd, const float *bn_g, const float *bn_b, float *out, hipStream_t s) {
    const int64_t n = (int64_t)B * T * d;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (kc == 9) hipLaunchKernelGGL(dwconv_bn_silu_kernel<9>, grid, dim3(256), 0, s, g, T, d, w, bias, bn_mean, bn_rstd, bn_g, bn_b, n, out);
    else if (kc == 31) hipLaunchKernelGGL(dwconv_bn_silu_kernel<31>, grid, dim3(256), 0, s, g, T, d, w, bias, bn_mean, bn_rstd, bn_g, bn_b, n, out);
}

}  // namespace pk
