// parakeet.cpp_amd/csrc/kernels/gemm_smallm.hip -- the fp32 MFMA GEMM for a HANDFUL of rows (M <= 64): the streaming
// encoder's products (16 streams x 1-3 frames per chunk against the full 600M-parameter weight set).
//
// out[M][N] = epi(A[M][K] * W[N][K]^T + bias), natural-k fma chains (bit-identical to the oracle and to the big-tile kernels).
// With so few rows the problem is a latency-bound weight stream, not a tile problem: the 64x64-tile kernel leaves N/64
// workgroups (16 for N = 1024) each walking a K = 4096 chain of 32x32x2 MFMAs -- 150 us.  Here:
//  * one WAVEFRONT per 16x16 output tile (v_mfma_f32_16x16x4_f32, 40-cycle dependent chain): N/16 x ceil(M/16) independent
//    workgroups (128 for N = 1024, M = 32), no barriers at all;
//  * K advances in chunks of 64: the lane (row, quarter) loads one float4 of A and of W per 16 k (coalesced 64-byte row
//    segments), parks it k-planar in LDS (plane k&3, so that the lane that feeds k = 4s + kq reads four consecutive steps with
//    one ds_read_b128 -- the same trick as the attention kernel), the loads of the next three chunks are in flight under the
//    MFMAs of chunk i;
//  * GLU keeps the value and the gate tile of the same 16 columns in one wave (two interleaved chains).
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

typedef float sm_f32x4 __attribute__((ext_vector_type(4)));

template <int EPI>
__global__ __launch_bounds__(64) void gemm_smallm_kernel(GemmArgs g) {
    constexpr int KC = 64;                       // k per chunk
    constexpr int PIT = KC / 4 + 4;              // plane row pitch (floats): 20 -> pitch/4 odd, conflict-free b128 reads
    constexpr int PLANE = 16 * PIT;
    constexpr int NB = (EPI == EPI_GLU) ? 2 : 1; // B tiles per wave (GLU: value + gate)
    __shared__ __attribute__((aligned(16))) float lds[(1 + NB) * 4 * PLANE];
    float *As = lds, *Ws = lds + 4 * PLANE;
    const int lane = threadIdx.x, r = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    // staging: lane (r, kq) moves float4 #kq of every 16-k block of row r
    int arow = m0 + r;
    arow = arow < g.M ? arow : g.M - 1;
    const float *ap = g.A + (int64_t)arow * g.lda + 4 * kq;
    const float *wp[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        int wrow = n0 + r;
        wrow = wrow < g.N ? wrow : g.N - 1;
        wp[b] = g.W + (int64_t)(b * g.N + wrow) * g.ldw + 4 * kq;
    }
    // register ring of DEPTH chunks: the loads of chunk i+DEPTH-1 are issued before chunk i is consumed, so ~DEPTH x 640 cycles of
    // MFMA chain cover the L2 / HBM latency of the weight stream (one chunk ahead left every iteration waiting ~1 us)
    constexpr int DEPTH = 4;
    float4 ra[DEPTH][4], rw[DEPTH][NB][4];
    auto gload = [&](int kc, int set) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[set][q] = *reinterpret_cast<const float4 *>(ap + kc * KC + 16 * q);
#pragma unroll
            for (int b = 0; b < NB; ++b) rw[set][b][q] = *reinterpret_cast<const float4 *>(wp[b] + kc * KC + 16 * q);
        }
    };
    auto park = [&](int set) {                   // element k = 16q + 4kq + e  ->  plane e, row r, column (k >> 2) = 4q + kq
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float *a = As + r * PIT + 4 * q + kq;
            a[0] = ra[set][q].x; a[PLANE] = ra[set][q].y; a[2 * PLANE] = ra[set][q].z; a[3 * PLANE] = ra[set][q].w;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float *w = Ws + b * 4 * PLANE + r * PIT + 4 * q + kq;
                w[0] = rw[set][b][q].x; w[PLANE] = rw[set][b][q].y; w[2 * PLANE] = rw[set][b][q].z; w[3 * PLANE] = rw[set][b][q].w;
            }
        }
    };
    sm_f32x4 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = sm_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int nkc = g.K / KC;
#pragma unroll
    for (int j = 0; j < DEPTH - 1; ++j)
        if (j < nkc) gload(j, j);
    for (int kc0 = 0; kc0 < nkc; kc0 += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const int kc = kc0 + j;
            if (kc < nkc) {
                __builtin_amdgcn_wave_barrier();     // the previous chunk's fragment reads are done (single wavefront: program order)
                park(j);
                if (kc + DEPTH - 1 < nkc) gload(kc + DEPTH - 1, (j + DEPTH - 1) % DEPTH);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // the lane that feeds k = 4s + kq reads plane kq: steps s = 4f .. 4f+3 per ds_read_b128
                const float *af = As + kq * PLANE + r * PIT;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const float4 a = *reinterpret_cast<const float4 *>(af + 4 * f);
                    float4 w[NB];
#pragma unroll
                    for (int b = 0; b < NB; ++b) w[b] = *reinterpret_cast<const float4 *>(Ws + b * 4 * PLANE + kq * PLANE + r * PIT + 4 * f);
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w[b].x, acc[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w[b].y, acc[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w[b].z, acc[b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w[b].w, acc[b], 0, 0, 0);
                }
            }
        }
    }
    // epilogue: C/D layout of 16x16x4: column = lane & 15, row = 4 * (lane >> 4) + i
    const int col = n0 + r;
    if (col >= g.N) return;
    const float bias = g.bias ? g.bias[col] : 0.0f;
    float bias_g = 0.0f;
    if constexpr (EPI == EPI_GLU) bias_g = g.bias ? g.bias[g.N + col] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + 4 * kq + i;
        if (row >= g.M) continue;
        float v = acc[0][i];
        if (g.bias) v = v + bias;
        if constexpr (EPI == EPI_RELU) {
            v = v > 0.0f ? v : 0.0f;
        } else if constexpr (EPI == EPI_SILU) {
            v = dsiluf(v);
        } else if constexpr (EPI == EPI_RESID) {
            const float y = v * g.alpha;
            v = g.resid[(int64_t)row * g.ldr + col] + y;
        } else if constexpr (EPI == EPI_GLU) {
            float gt = acc[NB - 1][i];
            if (g.bias) gt = gt + bias_g;
            v = v * dsigmoidf(gt);
        }
        if (g.remap_rows) g.out[(int64_t)(row / g.remap_rows) * g.remap_gs + (int64_t)(row % g.remap_rows) * g.remap_rs + (int64_t)col * g.remap_cs] = v;
        else g.out[(int64_t)row * g.ldo + (col < g.sigma_cols ? ((col & ~15) | ((col & 3) << 2) | ((col >> 2) & 3)) : col)] = v;
    }
}

void launch_gemm_smallm(const GemmArgs &a, int epi, hipStream_t s) {
    const dim3 grid((a.N + 15) / 16, (a.M + 15) / 16);
    switch (epi) {
    case EPI_NONE: hipLaunchKernelGGL(gemm_smallm_kernel<EPI_NONE>, grid, dim3(64), 0, s, a); break;
    case EPI_RELU: hipLaunchKernelGGL(gemm_smallm_kernel<EPI_RELU>, grid, dim3(64), 0, s, a); break;
    case EPI_SILU: hipLaunchKernelGGL(gemm_smallm_kernel<EPI_SILU>, grid, dim3(64), 0, s, a); break;
    case EPI_RESID: hipLaunchKernelGGL(gemm_smallm_kernel<EPI_RESID>, grid, dim3(64), 0, s, a); break;
    case EPI_GLU: hipLaunchKernelGGL(gemm_smallm_kernel<EPI_GLU>, grid, dim3(64), 0, s, a); break;
    default: break;
    }
}

}  // namespace pk
