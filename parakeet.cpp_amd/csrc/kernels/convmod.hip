// parakeet.cpp_amd/csrc/kernels/convmod.hip -- the middle of the Conformer convolution module
// (reference ConformerConvModule::forward, src/encoder.cpp:66-68): depthwise Conv1d(k, pad (k-1)/2,
// groups=d) -> BatchNorm1d (inference, running stats) -> SiLU, fused in one pass over [B][T][d].
// The pointwise convs + GLU on either side are epilogues of the MFMA GEMM.
// One thread = 4 adjacent channels x a strip of TT frames: the k input rows slide through registers (each row is loaded once
// per strip as a float4 instead of k times as scalars), taps / BatchNorm parameters stay in registers for the whole strip.
// Out-of-range taps multiply a zero row (fma(w, 0, acc) == acc: same bits as the oracle's skip-the-padding chain).
// BatchNorm is applied as written ((y-mean)*rstd*gamma+beta, rstd precomputed on the host) rather
// than folded into the taps, so the result is bit-identical to the oracle.
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

template <int KC, int TT>
__global__ __launch_bounds__(256) void dwconv_bn_silu_kernel(const float *__restrict__ g, int T, int d, int n_strips,
                                                             const float *__restrict__ w /*[KC][d]*/, const float *__restrict__ bias,
                                                             const float *__restrict__ bn_mean, const float *__restrict__ bn_rstd,
                                                             const float *__restrict__ bn_g, const float *__restrict__ bn_b,
                                                             int64_t n_items, float *__restrict__ out, int out_bf16, SeqRag rg) {
    constexpr int HALF = (KC - 1) / 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_items) return;
    const int d4 = d >> 2;
    const int c4 = (int)(idx % d4);
    int t0;
    int64_t row0;                                                       // first row of this utterance in the (packed) row axis
    if (rg.units.u) {                                                   // ragged batch (kernels.hpp: SeqRag): strips of TT frames of ONE utterance,
        const RagUnit un = rg.units.u[idx / d4];                        // zero padding at ITS first and last frame
        t0 = un.r0;
        T = rg.T[un.b];
        row0 = rg.T_off[un.b];
    } else {
        t0 = (int)((idx / d4) % n_strips) * TT;
        row0 = (idx / ((int64_t)d4 * n_strips)) * T;
    }
    const float4 *gp = reinterpret_cast<const float4 *>(g + row0 * d) + c4;       // row t at gp[t * d4]
    float4 *op = reinterpret_cast<float4 *>(out + row0 * d) + c4;
    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
    bf16x4_ *oph = reinterpret_cast<bf16x4_ *>(reinterpret_cast<__bf16 *>(out) + row0 * d) + c4;   // bf16 mode: the pw2 GEMM's operand (RNE)
    auto put = [&](int t, const float4 &v) {
        if (out_bf16 == 1) oph[(int64_t)t * d4] = bf16x4_{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        else if (out_bf16 == 2) {                                       // fp32, sigma K layout (GemmArgs::a_sigma): channel 16 b + 4 q + j -> 16 b + 4 j + q
            float *o = out + (row0 + t) * d + ((4 * c4) & ~15) + (c4 & 3);
            o[0] = v.x; o[4] = v.y; o[8] = v.z; o[12] = v.w;
        } else op[(int64_t)t * d4] = v;
    };
    float4 wt[KC];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) wt[kk] = reinterpret_cast<const float4 *>(w + (int64_t)kk * d)[c4];
    const float4 bi = reinterpret_cast<const float4 *>(bias)[c4], mu = reinterpret_cast<const float4 *>(bn_mean)[c4];
    const float4 rs = reinterpret_cast<const float4 *>(bn_rstd)[c4], ga = reinterpret_cast<const float4 *>(bn_g)[c4];
    const float4 be = reinterpret_cast<const float4 *>(bn_b)[c4];
    auto row = [&](int t) { return (t >= 0 && t < T) ? gp[(int64_t)t * d4] : make_float4(0.0f, 0.0f, 0.0f, 0.0f); };
    auto bn_silu = [&](float4 acc) {
        float4 v;
        v.x = __builtin_fmaf(((acc.x + bi.x) - mu.x) * rs.x, ga.x, be.x);
        v.y = __builtin_fmaf(((acc.y + bi.y) - mu.y) * rs.y, ga.y, be.y);
        v.z = __builtin_fmaf(((acc.z + bi.z) - mu.z) * rs.z, ga.z, be.z);
        v.w = __builtin_fmaf(((acc.w + bi.w) - mu.w) * rs.w, ga.w, be.w);
        if (out_bf16 == 1) {                                            // bf16 mode: tolerance-class activation (GemmArgs::fast_act)
            v.x = fast_siluf(v.x); v.y = fast_siluf(v.y); v.z = fast_siluf(v.z); v.w = fast_siluf(v.w);
        } else {
            float q[4] = {v.x, v.y, v.z, v.w};
            dsilu4(q);                                                  // the specification's values, short form (pk_devmath.h)
            v = make_float4(q[0], q[1], q[2], q[3]);
        }
        return v;
    };
    if constexpr (KC + TT - 1 <= 16) {
        // every input row of the strip is requested before the first is used: KC + TT - 1 independent 16-byte loads in flight per
        // thread (a load issued inside the frame loop stalled every frame for a full memory round trip: 34 us -> see DESIGN.md 8)
        float4 win[KC + TT - 1];                                        // win[i] = input row t0 + i - HALF
#pragma unroll
        for (int i = 0; i < KC + TT - 1; ++i) win[i] = row(t0 + i - HALF);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int t = t0 + tt;
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                acc.x = __builtin_fmaf(wt[kk].x, win[tt + kk].x, acc.x);
                acc.y = __builtin_fmaf(wt[kk].y, win[tt + kk].y, acc.y);
                acc.z = __builtin_fmaf(wt[kk].z, win[tt + kk].z, acc.z);
                acc.w = __builtin_fmaf(wt[kk].w, win[tt + kk].w, acc.w);
            }
            const float4 v = bn_silu(acc);
            if (t < T) put(t, v);
        }
    } else {
        float4 win[KC];                                                 // win[kk] = input row t + kk - HALF (sliding)
#pragma unroll
        for (int kk = 0; kk < KC - 1; ++kk) win[kk + 1] = row(t0 + kk - HALF);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int t = t0 + tt;
#pragma unroll
            for (int kk = 0; kk < KC - 1; ++kk) win[kk] = win[kk + 1];
            win[KC - 1] = row(t + HALF);
            if (t >= T) break;
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                acc.x = __builtin_fmaf(wt[kk].x, win[kk].x, acc.x);
                acc.y = __builtin_fmaf(wt[kk].y, win[kk].y, acc.y);
                acc.z = __builtin_fmaf(wt[kk].z, win[kk].z, acc.z);
                acc.w = __builtin_fmaf(wt[kk].w, win[kk].w, acc.w);
            }
            put(t, bn_silu(acc));
        }
    }
}

template <int TT>
static void launch_dwconv_tt(const float *g, int B, int T, int d, int kc, const float *w, const float *bias, const float *bn_mean,
                             const float *bn_rstd, const float *bn_g, const float *bn_b, float *out, hipStream_t s, int out_bf16, const SeqRag &rag) {
    const int n_strips = (T + TT - 1) / TT;
    const int64_t n_items = (rag.units.u ? (int64_t)rag.units.count : (int64_t)B * n_strips) * (d / 4);          // d % 4 == 0 (hidden sizes are multiples of 32)
    const dim3 grid((unsigned)((n_items + 255) / 256));
    if (kc == 9) hipLaunchKernelGGL((dwconv_bn_silu_kernel<9, TT>), grid, dim3(256), 0, s, g, T, d, n_strips, w, bias, bn_mean, bn_rstd, bn_g, bn_b, n_items, out, out_bf16, rag);
    else if (kc == 31) hipLaunchKernelGGL((dwconv_bn_silu_kernel<31, TT>), grid, dim3(256), 0, s, g, T, d, n_strips, w, bias, bn_mean, bn_rstd, bn_g, bn_b, n_items, out, out_bf16, rag);
}

void launch_dwconv_bn_silu(const float *g, int B, int T, int d, int kc, const float *w, const float *bias, const float *bn_mean,
                           const float *bn_rstd, const float *bn_g, const float *bn_b, float *out, hipStream_t s, int out_bf16, const SeqRag &rag) {
    // strips of 8 frames per thread amortise the window loads on large batches; a single utterance (the latency-bound small-batch route) has
    // only a handful of workgroups that way -- strips of 2 frames: four times the threads, a quarter of the serial work each
    // (ragged batch: B * T = the total number of packed rows, and rag.units holds strips of dwconv_strip_frames(that total) frames)
    if (dwconv_strip_frames((int64_t)B * T) == 2) launch_dwconv_tt<2>(g, B, T, d, kc, w, bias, bn_mean, bn_rstd, bn_g, bn_b, out, s, out_bf16, rag);
    else launch_dwconv_tt<8>(g, B, T, d, kc, w, bias, bn_mean, bn_rstd, bn_g, bn_b, out, s, out_bf16, rag);
}
int dwconv_strip_frames(int64_t total_rows) { return total_rows <= 2048 ? 2 : 8; }

}  // namespace pk
