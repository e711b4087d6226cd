// parakeet.cpp_amd/csrc/kernels/attention.hip -- relative-position multi-head attention core
// (reference ConformerAttention::rel_position_attention, src/encoder.cpp:135-171, with rel_shift
// :85-109 applied in closed form):
//     S[i][j] = ( (q_i + u_h) . k_j  +  (q_i + v_h) . P_h[j - i + T - 1] ) / sqrt(hd)
//     ctx_i   = softmax_j(S[i][:]) V
// One workgroup per (utterance, head, block of 64 query rows).  All three contractions run on the
// fp32 MFMA (32x32x2, natural-k chains); the [rows][T] score block lives in LDS only, the
// [B][H][T][2T-1] position-score tensor the reference materialises is never formed: each 32x32
// position tile is scattered straight to its shifted column j = p - (T-1) + i.
// Softmax is one wavefront per row with the canonical max / sum64 butterflies.
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
static constexpr int RB = 64;   // query rows per workgroup

__global__ __launch_bounds__(256) void relpos_attention_kernel(const float *__restrict__ qkv, int ldq, int d, int T, int hd,
                                                               const float *__restrict__ pos /*[2T-1][d]*/,
                                                               const float *__restrict__ bias_u, const float *__restrict__ bias_v,
                                                               float scale, float *__restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) float S[];   // [RB][ldS]
    const int H = d / hd;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int i0 = blockIdx.y * RB;
    const int rows = (T - i0) < RB ? (T - i0) : RB;
    const int ldS = T + 1 + ((T & 1) ? 0 : 0);                  // pitch: T+1 is odd for even T -> conflict-free column reads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const float *qb = qkv + (int64_t)b * T * ldq + h * hd;       // q rows of this (b,h)
    const float *kb = qb + d, *vb = qb + 2 * d;
    const float *pb = pos + h * hd;
    const int n_rt = (rows + 31) / 32, n_ct = (T + 31) / 32;

    // ---- phase 1a: content scores (q+u) K^T -> S -----------------------------------------------------------
    for (int tile = wave; tile < n_rt * n_ct; tile += 4) {
        const int ti = tile / n_ct, tj = tile % n_ct;
        int qi = i0 + ti * 32 + l31; qi = qi < T ? qi : T - 1;
        int kj = tj * 32 + l31;      kj = kj < T ? kj : T - 1;
        const float *qr = qb + (int64_t)qi * ldq + lh, *kr = kb + (int64_t)kj * ldq + lh;
        const float *ur = bias_u + h * hd + lh;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 8
        for (int s = 0; s < hd / 2; ++s) {
            const float a = qr[2 * s] + ur[2 * s];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, kr[2 * s], acc, 0, 0, 0);
        }
        const int j = tj * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int il = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (il < rows && j < T) S[il * ldS + j] = acc[r];
        }
    }
    __syncthreads();
    // ---- phase 1b: position scores (q+v) P^T, shifted, combined and scaled ----------------------------------
    // rows [i0+32ti, +32) x all j need p = j - i + T - 1 in [T-1-(i_hi), 2T-2-i_lo]
    {
        int n_items = 0;
        int first_pt[2], count_pt[2];
        for (int ti = 0; ti < n_rt; ++ti) {
            const int ilo = i0 + ti * 32, ihi = (ilo + 31) < (T - 1) ? (ilo + 31) : (T - 1);
            const int pmin = T - 1 - ihi, pmax = 2 * T - 2 - ilo;
            first_pt[ti] = pmin / 32;
            count_pt[ti] = pmax / 32 - pmin / 32 + 1;
            n_items += count_pt[ti];
        }
        for (int item = wave; item < n_items; item += 4) {
            int ti = 0, rem = item;
            if (n_rt > 1 && rem >= count_pt[0]) { ti = 1; rem -= count_pt[0]; }
            const int tp = first_pt[ti] + rem;
            int qi = i0 + ti * 32 + l31; qi = qi < T ? qi : T - 1;
            int pp = tp * 32 + l31;      pp = pp < 2 * T - 1 ? pp : 2 * T - 2;
            const float *qr = qb + (int64_t)qi * ldq + lh, *pr = pb + (int64_t)pp * d + lh;
            const float *vr = bias_v + h * hd + lh;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 8
            for (int s = 0; s < hd / 2; ++s) {
                const float a = qr[2 * s] + vr[2 * s];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pr[2 * s], acc, 0, 0, 0);
            }
            const int p = tp * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int j = p - (T - 1) + (i0 + il);
                if (il < rows && p < 2 * T - 1 && j >= 0 && j < T) {
                    const float c = S[il * ldS + j];
                    S[il * ldS + j] = (c + acc[r]) * scale;      // (content + pos) * scale, src/encoder.cpp:157-160
                }
            }
        }
    }
    __syncthreads();
    // ---- phase 2: softmax, one wavefront per row -----------------------------------------------------------
    for (int il = wave; il < rows; il += 4) {
        float *row = S + il * ldS;
        float m = -__builtin_huge_valf();
        for (int j = lane; j < T; j += 64) m = fmaxf(m, row[j]);
        m = wave_max64(m);
        float p = 0.0f;
        for (int j = lane; j < T; j += 64) {
            const float e = dexpf(row[j] - m);
            row[j] = e;
            p = p + e;
        }
        const float sum = wave_sum64(p);
        for (int j = lane; j < T; j += 64) row[j] = row[j] / sum;
    }
    __syncthreads();
    // ---- phase 3: ctx = softmax(S) V  (k = key index, natural order) ---------------------------------------
    const int n_dt = hd / 32;
    for (int tile = wave; tile < n_rt * n_dt; tile += 4) {
        const int ti = tile / n_dt, td = tile % n_dt;
        int il = ti * 32 + l31; il = il < rows ? il : rows - 1;
        const float *sr = S + il * ldS;
        const float *vc = vb + td * 32 + l31;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const int steps = (T + 1) / 2;
#pragma unroll 4
        for (int s = 0; s < steps; ++s) {
            const int j = 2 * s + lh;
            const bool ok = j < T;                                // odd T: the last k of the last step is a zero pad
            const float a = ok ? sr[j] : 0.0f;
            const float v = vc[(int64_t)(ok ? j : T - 1) * ldq];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ir = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (ir < rows) ctx[((int64_t)b * T + i0 + ir) * d + h * hd + td * 32 + l31] = acc[r];
        }
    }
}

void launch_relpos_attention(const float *qkv, int B, int T, int d, int n_heads, const float *pos, const float *bias_u,
                             const float *bias_v, float *ctx, hipStream_t s) {
    const int hd = d / n_heads;
    const float scale = 1.0f / sqrtf((float)hd);                 // src/encoder.cpp:126
    const size_t lds = (size_t)RB * (T + 1) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&relpos_attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    dim3 grid(B * n_heads, (T + RB - 1) / RB);
    hipLaunchKernelGGL(relpos_attention_kernel, grid, dim3(256), lds, s, qkv, 3 * d, d, T, hd, pos, bias_u, bias_v, scale, ctx);
}

}  // namespace pk
