// tools/experiments/kernels/attention_t128.hip -- EXPERIMENT of round 3, not part of the library (kept so the measurement can be repeated: copy it into
// parakeet.cpp_amd/csrc/kernels/, declare the two entry points in kernels.hpp and route run_layers / ensure_pos_tables to it with NATURAL q / k / P
// columns).  Result on MI355X, tdt-ctc-110m 64 x 10 s (T = 126): bit-identical to the oracle on the whole encoder / e2e suite at the first attempt,
// and SLOWER than the general kernel: 86 us per layer (95 us with a 3-deep operand ring: spills) against 76 us.  2048 long wavefronts = exactly one
// round at two per SIMD expose every L2 round trip of the K / P operand stream (36 chunks of 8 MFMAs per wave); the general kernel runs four times as
// many, four times shorter wavefronts at 3-4 per SIMD and hides them.  What it would need: K / P tiles shared by the four waves through LDS (one
// cooperative fill per tile) and 16-row query tiles (8 waves per (utterance, head)).
//
// bit-exact fp32 relative-position attention for short sequences (T <= 128 encoder frames =
// clips of up to ~10.2 s: the shape of the headline benchmark, 64 x 10 s -> T = 126), head size 64.
// Reference: ConformerAttention::rel_position_attention (src/encoder.cpp:135-171) with rel_shift (:85-109) in closed form:
//     S[i][j] = ( (q_i + u_h) . k_j  +  (q_i + v_h) . P_h[j - i + T - 1] ) / sqrt(hd) ,   ctx_i = softmax_j(S[i][:]) V
// Same bits as attention.hip (the general kernel: 16x16x4 MFMA, [32][T] score block in LDS) and as the oracle; a different shape of the work,
// carried over from the bf16 kernel of round 3 (attention_bf16.hip):
//   * one workgroup per (utterance, head), four wavefronts x 32 query rows; every product is formed TRANSPOSED (keys x queries) on
//     v_mfma_f32_32x32x2_f32 -- bit-for-bit a k-ordered fmaf chain (lanes 0-31 feed k = 2s, lanes 32-63 k = 2s+1), the oracle's order.  In the
//     32x32 accumulator layout a lane holds 16 keys of ONE query, so the whole score row of a query (<= 128 keys = 4 tiles = 64 registers in
//     two lanes) stays in REGISTERS: no score block in LDS, no workgroup barriers in the score / softmax phases.
//   * rel_shift through a wave-private LDS strip exactly as in the bf16 kernel: one new 32 x 32 block of (q + v) P^T per key tile, skewed
//     conflict-free read, (content + position) * scale in the reference's operation order.
//   * softmax: the row maximum is an in-lane reduction + one exchange with lane ^ 32; exp is the contract's fixed polynomial; the
//     denominator is the CANONICAL sum64 (64 strided partial sums in increasing index, then the xor butterfly 32, 16, 8, 4, 2, 1) evaluated
//     on the register layout: key j = 32 t + jj with jj = (r & 3) + 8 (r >> 2) + 4 g, so canonical lane jj holds e[t=0][r] + e[t=2][r], lane
//     32 + jj holds e[t=1][r] + e[t=3][r], butterfly stages 32 / 16 / 8 / 2 / 1 are in-lane adds over r and stage 4 is the lane ^ 32 exchange
//     -- the same expression tree, the same bits.  Then one IEEE division per element, as the reference.
//   * softmax(S) V: the MFMA's k operand of lane half g must be key 2s + g, the registers hold the keys with bit 2 = g: half of them are
//     exchanged with lane ^ 32 (8 per tile), then the chain runs over the keys in natural order; V^T comes from an LDS copy of V (32 KB).
// K and P rows are read in their natural layout (16-byte loads, the lane keeps the even or the odd k), one chunk of 16 k ahead of the MFMAs.
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

typedef float ax_f32x16 __attribute__((ext_vector_type(16)));
static constexpr int AX_SKP = 34;       // skew strip pitch (floats)
static constexpr int AX_HD = 64;

__device__ __forceinline__ int ax_rowidx(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

__global__ __launch_bounds__(256, 2) void relpos_attention_t128_kernel(const float *__restrict__ qkv, int ldq, int d, int T,
                                                                       const float *__restrict__ pos /*[2T-1][d], natural columns*/,
                                                                       const float *__restrict__ bias_u, const float *__restrict__ bias_v, float scale,
                                                                       float *__restrict__ ctx, int n_bh) {
    constexpr int HD = AX_HD, NS = HD / 2;                          // MFMA k-steps of a contraction over the head dimension
    extern __shared__ __attribute__((aligned(16))) float ax_smem[];
    float *Vs = ax_smem;                                            // [128][HD]  V of this (utterance, head); rows >= T are zero
    float *skew = ax_smem + 128 * HD;                               // [4][64 * AX_SKP]
    __builtin_amdgcn_s_setprio(3);
    const int bh = blockIdx.x;
    if (bh >= n_bh) return;
    const int H = d / HD, P = 2 * T - 1;
    const int b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, g = lane >> 5;
    const float *qb = qkv + (int64_t)b * T * ldq + h * HD;          // q rows of (b, h); k at + d, v at + 2 d
    const float *kb = qb + d, *vb = qb + 2 * d;
    const float *pb = pos + h * HD;
    for (int e = tid; e < 128 * (HD / 4); e += 256) {               // V -> LDS (coalesced 16-byte loads)
        const int row = e / (HD / 4), c4 = e % (HD / 4);
        const float4 v = row < T ? *reinterpret_cast<const float4 *>(vb + (int64_t)row * ldq + 4 * c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *reinterpret_cast<float4 *>(Vs + row * HD + 4 * c4) = v;
    }
    __syncthreads();
    const int i0w = 32 * wave;
    if (i0w >= T) return;                                           // (no barrier below)
    float *sk = skew + wave * 64 * AX_SKP;
    const int base0 = T - 32 - i0w;                                 // p of band row 0 of block 0

    // lane (n, g): operand element of step s is k = 2 s + g.  A row's 16 k of chunk c arrive as four float4; the lane keeps .x/.z (g = 0) or .y/.w.
    float qu[NS], qv[NS];
    {
        int qr = i0w + n;
        qr = qr < T ? qr : T - 1;
        const float4 *qp = reinterpret_cast<const float4 *>(qb + (int64_t)qr * ldq);
        const float4 *up = reinterpret_cast<const float4 *>(bias_u + h * HD), *vp = reinterpret_cast<const float4 *>(bias_v + h * HD);
#pragma unroll
        for (int j = 0; j < HD / 4; ++j) {
            const float4 f = qp[j], u4 = up[j], v4 = vp[j];
            const float q0 = g ? f.y : f.x, q1 = g ? f.w : f.z;
            qu[2 * j] = q0 + (g ? u4.y : u4.x);                     // (q + u), (q + v) as the reference forms them (src/encoder.cpp:141-142)
            qu[2 * j + 1] = q1 + (g ? u4.w : u4.z);
            qv[2 * j] = q0 + (g ? v4.y : v4.x);
            qv[2 * j + 1] = q1 + (g ? v4.w : v4.z);
        }
    }
    // the nine A-operand row sets in the order they are consumed: K tiles 0..3, then P blocks 0..4
    auto rowptr = [&](int idx) -> const float4 * {
        if (idx < 4) {
            int kr = 32 * idx + n;
            kr = kr < T ? kr : T - 1;
            return reinterpret_cast<const float4 *>(kb + (int64_t)kr * ldq);
        }
        int pr = base0 + 32 * (idx - 4) + n;
        pr = pr < 0 ? 0 : (pr > P - 1 ? P - 1 : pr);
        return reinterpret_cast<const float4 *>(pb + (int64_t)pr * d);
    };
    // operand chunks in a ring of AX_PF slots: chunk q + AX_PF - 1 is requested before the MFMAs of chunk q (two waves per SIMD cannot hide an L2
    // round trip per 8 MFMAs by themselves)
    constexpr int PF = 3;
    float4 ring[PF][4];
    auto issue = [&](int q, int slot) {                             // chunk q = (row set q >> 2, 16 k at 16 (q & 3))
        const float4 *p = rowptr(q >> 2) + 4 * (q & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) ring[slot][j] = p[j];
    };
    ax_f32x16 S[4];
#pragma unroll
    for (int q = 0; q < PF - 1; ++q) issue(q, q);
#pragma unroll
    for (int idx = 0; idx < 9; ++idx) {
        ax_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            constexpr int dummy = 0; (void)dummy;
            const int q = idx * 4 + c, slot = q % PF;
            if (q + PF - 1 < 36) issue(q + PF - 1, (q + PF - 1) % PF);   // flies under the MFMAs of this chunk and the next
            float a[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[2 * j] = g ? ring[slot][j].y : ring[slot][j].x;
                a[2 * j + 1] = g ? ring[slot][j].w : ring[slot][j].z;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], idx < 4 ? qu[8 * c + e] : qv[8 * c + e], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (idx < 4) {
            S[idx] = acc;                                           // content scores of key tile idx (transposed: [key][query])
        } else {
            const int m = idx - 4;                                  // position block m -> strip half m & 1
#pragma unroll
            for (int r = 0; r < 16; ++r) sk[(32 * (m & 1) + ax_rowidx(r, g)) * AX_SKP + n] = acc[r];
            if (m >= 1) {                                           // key tile t = m - 1 has both of its blocks: skewed read, combine, scale
                const int t = m - 1;
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jj = ax_rowidx(r, g);
                    const int xr = jj - n + 31;                      // band row relative to block t: 0 .. 62
                    const int phys = 32 * ((t + (xr >> 5)) & 1) + (xr & 31);
                    S[t][r] = (S[t][r] + sk[phys * AX_SKP + n]) * scale;      // (content + position) * scale, src/encoder.cpp:157-160
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // block t's half is overwritten by block t + 2
            }
        }
    }
    // ---- softmax over the query's row (two lanes x 4 tiles x 16 registers) ----
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (32 * t + ax_rowidx(r, g) < T) mx = fmaxf(mx, S[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[t][r] = (32 * t + ax_rowidx(r, g) < T) ? dexpf_nonpos(S[t][r] - mx) : 0.0f;   // S <= row maximum
    float sum;
    {   // the canonical sum64 on this layout (see the header): stages 32, 16, 8 in-lane, 4 = lane ^ 32, then 2, 1 in-lane
        float c[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = (S[0][r] + S[2][r]) + (S[1][r] + S[3][r]);
        float d8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) d8[r] = c[r] + c[r + 8];
        float f4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) f4[r] = d8[r] + d8[r + 4];
#pragma unroll
        for (int r = 0; r < 4; ++r) f4[r] = f4[r] + __shfl_xor(f4[r], 32, 64);
        sum = (f4[0] + f4[2]) + (f4[1] + f4[3]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[t][r] = S[t][r] / sum;
    // ---- ctx^T += V^T P^T over the keys in natural order ----
    ax_f32x16 O[HD / 32];
#pragma unroll
    for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[dt][r] = 0.0f;
    const int nsteps = (T + 1) / 2;                                 // MFMA steps (2 keys each) that hold at least one key of the row
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        // lane half g needs the keys of parity g; its registers hold the keys with bit 2 = g: send the other parity's to lane ^ 32
        float recv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) recv[k] = __shfl_xor(g ? S[t][2 * k] : S[t][2 * k + 1], 32, 64);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (16 * t + s < nsteps) {                              // (wave-uniform)
                const int kk0 = 2 * s, kk1 = 2 * s + 1;
                const int r0 = (kk0 & 3) + 4 * (kk0 >> 3), r1 = (kk1 & 3) + 4 * (kk1 >> 3);
                const float v0 = ((kk0 >> 2) & 1) == 0 ? S[t][r0] : recv[r0 >> 1];
                const float v1 = ((kk1 >> 2) & 1) == 1 ? S[t][r1] : recv[r1 >> 1];
                const float pB = g ? v1 : v0;
                const float *vrow = Vs + (32 * t + 2 * s + g) * HD + n;
#pragma unroll
                for (int dt = 0; dt < HD / 32; ++dt) O[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * dt], pB, O[dt], 0, 0, 0);
            }
        }
    }
    const int i = i0w + n;
    if (i < T) {
        float *orow = ctx + ((int64_t)b * T + i) * d + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
                *reinterpret_cast<float4 *>(orow + 32 * dt + 8 * rq) = make_float4(O[dt][4 * rq], O[dt][4 * rq + 1], O[dt][4 * rq + 2], O[dt][4 * rq + 3]);
    }
}

bool relpos_attention_t128_applies(int T, int hd) { return hd == AX_HD && T >= 1 && T <= 128; }

void launch_relpos_attention_t128(const float *qkv, int B, int T, int d, int n_heads, const float *pos, const float *bias_u, const float *bias_v,
                                  float *ctx, hipStream_t s) {
    const float scale = 1.0f / sqrtf((float)AX_HD);                // src/encoder.cpp:126
    const size_t lds = (size_t)(128 * AX_HD + 4 * 64 * AX_SKP) * sizeof(float);
    static DynLdsSlots slots;
    ensure_dyn_lds(slots, reinterpret_cast<const void *>(&relpos_attention_t128_kernel), lds);
    hipLaunchKernelGGL(relpos_attention_t128_kernel, dim3(B * n_heads), dim3(256), lds, s, qkv, 3 * d, d, T, pos, bias_u, bias_v, scale, ctx, B * n_heads);
}

}  // namespace pk
