// parakeet.cpp_amd/csrc/kernels/decode_gemv.hip -- the three per-step products of the TDT / RNNT loop
// (reference: LSTMCell::forward src/lstm.cpp:11-29, TDTJoint::forward src/tdt.cpp:15-24) for a lock-step batch
// of B <= a few hundred utterances:   out[B][N] = X[B][K] * W[N][K]^T,  K = 640.
//
// This is a latency problem (a 640-long dependent fp32 chain per output, ~130 dependent steps per clip), not a
// throughput one, so the tiling differs from the big GEMM:
//  * v_mfma_f32_16x16x4_f32 (32-cycle issue, 40-cycle dependent latency): one 16x16 (utterance x output) tile per
//    wavefront, a single accumulator chain of K/4 MFMAs in natural k order -> bit-identical to the oracle.
//  * Operands come straight from L2 with 16-byte loads: both X and W are stored in a "sigma" K layout -- inside every
//    block of 16 k the 4x4 index matrix is transposed -- so the float4 a lane loads holds exactly its k = 4s+kq
//    values for four consecutive MFMA steps.  W is permuted once at upload; X (h, z) is *written* permuted by the
//    fused epilogues of the previous product.
//  * A workgroup = one 16-output tile x up to four 16-utterance tiles: the W tile is fetched from L2 once and hits in
//    L1 for the other three waves; the per-XCD slice of the 10.8 MB of decode weights stays L2-resident across steps.
//  * Epilogues: LSTM cell (gates -> c', h'), joint activation relu(enc_proj[t_b] + .), or bias.
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int sigma16(int k) { return (k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3); }

// NCH: compile-time number of 64-wide K chunks (10 for K = 640: fully unrolled, counted vmcnt waits keep the next
// chunk's loads in flight under the MFMA chain); 0 = runtime trip count (any K % 64 == 0).
template <int EPI, int NCH>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyArgs a) {
    __shared__ float tile[4][16][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int nt = blockIdx.x;
    const int m0 = (blockIdx.y * 4 + wave) * 16;
    if (m0 >= a.B) return;                                       // whole wave out of range (uniform)
    int wrow;
    if (EPI == SK_CELL) wrow = (col >> 2) * a.Hp + 4 * nt + (col & 3);   // tile columns = (gate, unit): rows g*Hp + j
    else { wrow = 16 * nt + col; wrow = wrow < a.N ? wrow : a.N - 1; }
    int xrow = m0 + col;
    xrow = xrow < a.B ? xrow : a.B - 1;
    const float4 *xp = reinterpret_cast<const float4 *>(a.X + (int64_t)xrow * a.K) + kq;
    const float4 *wp = reinterpret_cast<const float4 *>(a.W + (int64_t)wrow * a.K) + kq;
    // Epilogue operands are fetched FIRST (token -> g1 row, c, enc_proj[t_b], bias): in this latency-bound loop every
    // dependent round trip to L2 / HBM that can hide under the 160-MFMA chain is ~1-2 us saved per launch.
    float e_gi[4] = {0.0f, 0.0f, 0.0f, 0.0f}, e_c = 0.0f;          // SK_CELL: lane -> (utterance lane>>2, unit lane&3)
    float e_ep[4] = {0.0f, 0.0f, 0.0f, 0.0f}, e_bias = 0.0f;       // SK_ACT / SK_BIAS: lane -> column `col`, utterances 4*kq+r
    if (EPI == SK_CELL) {
        const int b = m0 + (lane >> 2), j = 4 * nt + (lane & 3);
        if (b < a.B) {
            const float *gir = a.gi + (int64_t)(a.gi_row ? a.gi_row[b] : b) * a.gi_ld;
#pragma unroll
            for (int g = 0; g < 4; ++g) e_gi[g] = gir[g * a.Hp + j];
            e_c = a.c[(int64_t)b * a.Hp + j];
        }
    } else {
        const int n = 16 * nt + col;
        if (n < a.N) {
            if (a.bias) e_bias = a.bias[n];
            if (EPI == SK_ACT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = m0 + 4 * kq + r;
                    if (b < a.B) {
                        int tt = a.t[b];
                        tt = tt < a.T ? tt : a.T - 1;
                        e_ep[r] = a.ep[((int64_t)b * a.T + tt) * a.N + n];
                    }
                }
            }
        }
    }
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    // software pipeline: chunks of 4 float4 pairs (16 MFMAs, ~640 cycles) with the next chunk's loads in flight
    constexpr int CH = 4;
    const int nchunks = a.K / (16 * CH);
    float4 xa[CH], wa[CH], xb[CH], wb[CH];
#define SK_LOAD(X_, W_, c_)                                                         \
    _Pragma("unroll") for (int i = 0; i < CH; ++i) {                               \
        X_[i] = xp[4 * ((c_) * CH + i)];                                            \
        W_[i] = wp[4 * ((c_) * CH + i)];                                            \
    }                                                                               \
    __builtin_amdgcn_sched_barrier(0);
#define SK_MMA(X_, W_)                                                              \
    _Pragma("unroll") for (int i = 0; i < CH; ++i) {                               \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(X_[i].x, W_[i].x, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(X_[i].y, W_[i].y, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(X_[i].z, W_[i].z, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(X_[i].w, W_[i].w, acc, 0, 0, 0); \
    }                                                                               \
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NCH > 0) {
        // two register sets, fully unrolled: chunk c+1 is in flight while chunk c feeds the MFMA chain (>= 640 cycles of cover).
        // A third set hid more latency but pushed the kernel past 96 VGPRs, and then a decode wave no longer fits next to the
        // four 104-VGPR waves per SIMD of the 128x128 GEMM of the NEXT batch's encoder (two-stream pipeline): every decode
        // workgroup had to wait for a GEMM workgroup to retire and then held that slot -- 2.0 ms per 64-clip batch.
        SK_LOAD(xa, wa, 0)
#pragma unroll
        for (int c = 0; c < NCH; c += 2) {
            if (c + 1 < NCH) { SK_LOAD(xb, wb, c + 1) }
            SK_MMA(xa, wa)
            if (c + 2 < NCH) { SK_LOAD(xa, wa, c + 2) }
            if (c + 1 < NCH) { SK_MMA(xb, wb) }
        }
    } else {
        SK_LOAD(xa, wa, 0)
        for (int c = 0; c < nchunks; c += 2) {
            if (c + 1 < nchunks) { SK_LOAD(xb, wb, c + 1) }
            SK_MMA(xa, wa)
            if (c + 2 < nchunks) { SK_LOAD(xa, wa, c + 2) }
            if (c + 1 < nchunks) { SK_MMA(xb, wb) }
        }
    }
#undef SK_LOAD
#undef SK_MMA
    // C/D layout of 16x16x4: column = lane & 15, row (utterance) = 4 * (lane >> 4) + r
    if (EPI == SK_BIAS) {
        const int n = 16 * nt + col;
        if (n < a.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = m0 + 4 * kq + r;
                if (b < a.B) a.out[(int64_t)b * a.ldo + n] = a.bias ? acc[r] + e_bias : acc[r];
            }
        }
    } else if (EPI == SK_ACT) {
        // z = relu(enc_proj(enc_t) + pred_proj(pred) [+ bp])   src/tdt.cpp:17-18 ; written in sigma layout
        const int n = 16 * nt + col;
        if (n < a.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = m0 + 4 * kq + r;
                if (b >= a.B) continue;
                float p = acc[r];
                if (a.bias) p = p + e_bias;
                const float s = e_ep[r] + p;
                a.out[(int64_t)b * a.N + sigma16(n)] = s > 0.0f ? s : 0.0f;
            }
        }
    } else {
        // LSTMCell::forward: gates = (W_ih x + b) + W_hh h ; i,f,g,o ; c' = f*c + i*g ; h' = o*tanh(c')
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[wave][4 * kq + r][col] = acc[r];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int ul = lane >> 2, jj = lane & 3;
        const int b = m0 + ul, j = 4 * nt + jj;
        if (b < a.B) {
            const float gi_ = e_gi[0] + tile[wave][ul][jj];
            const float gf_ = e_gi[1] + tile[wave][ul][4 + jj];
            const float gg_ = e_gi[2] + tile[wave][ul][8 + jj];
            const float go_ = e_gi[3] + tile[wave][ul][12 + jj];
            const float ig = dsigmoidf(gi_), fg = dsigmoidf(gf_), gg = dtanhf(gg_), og = dsigmoidf(go_);
            const float t1 = fg * e_c;
            const float t2 = ig * gg;
            const float cnew = t1 + t2;
            a.cn[(int64_t)b * a.Hp + j] = cnew;
            a.out[(int64_t)b * a.Hp + sigma16(j)] = og * dtanhf(cnew);   // h' in sigma layout (it is only ever a GEMV operand)
        }
    }
}

void launch_skinny_gemm(const SkinnyArgs &a, int epi, hipStream_t s) {
    const int n_tiles = epi == SK_CELL ? a.Hp / 4 : (a.N + 15) / 16;
    dim3 grid(n_tiles, (a.B + 63) / 64);
    const bool k640 = a.K == 640;
    switch (epi) {
    case SK_BIAS:
        if (k640) hipLaunchKernelGGL((skinny_gemm_kernel<SK_BIAS, 10>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((skinny_gemm_kernel<SK_BIAS, 0>), grid, dim3(256), 0, s, a);
        break;
    case SK_ACT:
        if (k640) hipLaunchKernelGGL((skinny_gemm_kernel<SK_ACT, 10>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((skinny_gemm_kernel<SK_ACT, 0>), grid, dim3(256), 0, s, a);
        break;
    case SK_CELL:
        if (k640) hipLaunchKernelGGL((skinny_gemm_kernel<SK_CELL, 10>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((skinny_gemm_kernel<SK_CELL, 0>), grid, dim3(256), 0, s, a);
        break;
    default: break;
    }
}

}  // namespace pk
