// parakeet.cpp_amd/csrc/kernels/attention.hip -- relative-position multi-head attention core
// (reference ConformerAttention::rel_position_attention, src/encoder.cpp:135-171, with rel_shift
// :85-109 applied in closed form):
//     S[i][j] = ( (q_i + u_h) . k_j  +  (q_i + v_h) . P_h[j - i + T - 1] ) / sqrt(hd)
//     ctx_i   = softmax_j(S[i][:]) V
// One workgroup per (utterance, head, block of 32 query rows), 4 wavefronts.  All three contractions run on the
// fp32 MFMA v_mfma_f32_16x16x4_f32 (natural-k fma chains = the oracle's order); 16x16 tiles give every wave the same
// number of tiles in each phase.  Everything an MFMA consumes is staged through LDS with coalesced 16-byte global loads:
//   * Q (32 rows) once, then held in registers as the A fragments of q+u and q+v;
//   * K, the needed band of P_h (T+31 rows: p = j - i + T - 1), and V stream through ONE chunk buffer of CH rows, so the
//     footprint does not grow with T (2 workgroups per CU at hd = 64); the rows of chunk n+1 are in flight in registers
//     while chunk n feeds the MFMAs;
//   * the [32][T] score block lives in LDS only.  The [B][H][T][2T-1] position-score tensor the reference materialises
//     is never formed: each 16x16 position tile is added straight into its shifted column j = p - (T-1) + i.
// LDS layouts are "k-planar": element (row, k) of a K-contiguous operand sits at plane (k&3), row*pitch + (k>>2), so the
// lane that feeds k = 4s+kq reads its operands of four consecutive MFMA steps with one conflict-free ds_read_b128
// (pitch/4 odd).  The score block uses the same layout over the key index j (it is the A operand of softmax(S) V).
// Softmax is one wavefront per row with the canonical max / sum64 butterflies.
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
static constexpr int RB = 32;   // query rows per workgroup

__device__ __forceinline__ float f4e(const float4 &v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

template <int HD, int CH>
__global__ __launch_bounds__(256) void relpos_attention_kernel(const float *__restrict__ qkv, int ldq, int d, int T,
                                                               const float *__restrict__ pos /*[2T-1][d]*/,
                                                               const float *__restrict__ bias_u, const float *__restrict__ bias_v,
                                                               float scale, float *__restrict__ ctx, int PITS, int s_floats, int n_rb, int n_bh) {
    constexpr int KQ = HD / 4;            // k-steps of a full head-dim contraction = floats per plane row
    constexpr int PITQ = KQ + 4;          // plane row pitch of K / P / Q tiles (PITQ/4 odd)
    constexpr int NQ4 = HD / 16;          // float4 fragments per lane for K = HD
    constexpr int VPIT = HD + 16;         // V rows, natural layout (pitch = 16 mod 32 banks)
    constexpr int NDV = HD / 32;          // 16-wide ctx column tiles per wave (HD/16 tiles over 2 column parities)
    constexpr int QPLANE = RB * PITQ, KPLANE = CH * PITQ;
    constexpr int NLD = CH * KQ / 256, NLQ = RB * KQ / 256;         // float4 loads per thread: one chunk / the Q tile
    static_assert((CH * KQ) % 256 == 0 && (RB * KQ) % 256 == 0 && CH % 16 == 0, "chunk must split evenly over 256 threads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *S = smem;                      // [4][RB][PITS] score planes; first holds the Q tile [4][RB][PITQ]
    float *KP = smem + s_floats;          // [4][CH][PITQ] K / P chunk, or [CH][VPIT] V chunk
    const int SPLANE = RB * PITS;
    const int H = d / HD;
    // Block b runs on XCD b % 8 (observed dispatch rule).  The row blocks of one (utterance, head) share K, V and the P band:
    // give them consecutive slots on ONE XCD so the second..last read those rows from that XCD's L2 instead of HBM.
    int bh, rbk;
    {
        const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
        rbk = k % n_rb;
        bh = (k / n_rb) * 8 + xcd;
        if (bh >= n_bh) return;                                      // padding slot of the grid (whole workgroup, before any barrier)
    }
    const int b = bh / H, h = bh % H;
    const int i0 = rbk * RB;
    const int rows = (T - i0) < RB ? (T - i0) : RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int rt = wave & 1, cp = wave >> 1;                       // this wave's 16-row tile and tile-column parity
    const int P = 2 * T - 1;
    const float *qb = qkv + (int64_t)b * T * ldq + h * HD;          // q rows of this (b,h)
    const float *kb = qb + d, *vb = qb + 2 * d;
    const float *pb = pos + h * HD;
    auto sidx = [&](int il, int j) { return (j & 3) * SPLANE + il * PITS + (j >> 2); };

    // ---- the stream of chunk jobs: K chunks, then the needed band of P, then V chunks ------------------------------------
    // A job's rows are fetched (coalesced 16-byte loads, HD/4 consecutive threads per row) into registers while the
    // previous job computes, and committed to the chunk buffer between two barriers.
    const int i_hi = (i0 + RB - 1) < (T - 1) ? (i0 + RB - 1) : (T - 1);
    const int pmin = T - 1 - i_hi, pmax = 2 * T - 2 - i0;           // band of P rows this block needs
    const int p_first = pmin;                                        // tiles are relative to the chunk: no alignment needed
    const int nK = (T + CH - 1) / CH, nP = (pmax - p_first) / CH + 1, njobs = 2 * nK + nP;
    float4 pf[NLD];
    auto job_src = [&](int job, const float *&src, int64_t &ld, int &row0, int &limit) {
        if (job < nK) { src = kb; ld = ldq; row0 = job * CH; limit = T; }
        else if (job < nK + nP) { src = pb; ld = d; row0 = p_first + (job - nK) * CH; limit = P; }
        else { src = vb; ld = ldq; row0 = (job - nK - nP) * CH; limit = T; }
    };
    auto issue = [&](int job) {
        const float *src; int64_t ld; int row0, limit;
        job_src(job, src, ld, row0, limit);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + 256 * i, gr = row0 + e / KQ;
            pf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (gr < limit) pf[i] = *reinterpret_cast<const float4 *>(src + (int64_t)gr * ld + 4 * (e % KQ));
        }
    };
    auto commit = [&](int job) {
        if (job < nK + nP) {                                       // K / P: k-planar
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = tid + 256 * i;
                float *q = KP + (e / KQ) * PITQ + (e % KQ);
                q[0] = pf[i].x; q[KPLANE] = pf[i].y; q[2 * KPLANE] = pf[i].z; q[3 * KPLANE] = pf[i].w;
            }
        } else {                                                    // V: natural rows
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = tid + 256 * i;
                *reinterpret_cast<float4 *>(KP + (e / KQ) * VPIT + 4 * (e % KQ)) = pf[i];
            }
        }
    };

    // ---- phase 0: Q tile -> registers as the A fragments of (q+u) and (q+v) ---------------------------------------------
    {
        float4 qf[NLQ];
#pragma unroll
        for (int i = 0; i < NLQ; ++i) {
            const int e = tid + 256 * i, gr = i0 + e / KQ;
            qf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (gr < T) qf[i] = *reinterpret_cast<const float4 *>(qb + (int64_t)gr * ldq + 4 * (e % KQ));
        }
        issue(0);
#pragma unroll
        for (int i = 0; i < NLQ; ++i) {
            const int e = tid + 256 * i;
            float *q = S + (e / KQ) * PITQ + (e % KQ);
            q[0] = qf[i].x; q[QPLANE] = qf[i].y; q[2 * QPLANE] = qf[i].z; q[3 * QPLANE] = qf[i].w;
        }
        commit(0);
    }
    __syncthreads();
    float4 qu[NQ4], qv[NQ4];
    {
        const float *qp = S + kq * QPLANE + (rt * 16 + l15) * PITQ;
        const float *ur = bias_u + h * HD + kq, *vr = bias_v + h * HD + kq;
#pragma unroll
        for (int f = 0; f < NQ4; ++f) {
            const float4 q = *reinterpret_cast<const float4 *>(qp + 4 * f);
            // k = 4*(4f+e) + kq ; (q+u), (q+v) as the reference forms them (src/encoder.cpp:141-142)
            qu[f] = make_float4(q.x + ur[16 * f], q.y + ur[16 * f + 4], q.z + ur[16 * f + 8], q.w + ur[16 * f + 12]);
            qv[f] = make_float4(q.x + vr[16 * f], q.y + vr[16 * f + 4], q.z + vr[16 * f + 8], q.w + vr[16 * f + 12]);
        }
    }
    __syncthreads();                                              // the Q tile's space becomes the score block

    // one or two 16x16 tiles of A(q-fragments) x B(rows of the chunk buffer)^T, K = HD
    auto tile_pair = [&](const float4 (&a)[NQ4], int t0, int t1, bool two, f32x4 &c0, f32x4 &c1) {
        const float *b0 = KP + kq * KPLANE + (t0 * 16 + l15) * PITQ;
        const float *b1 = KP + kq * KPLANE + (t1 * 16 + l15) * PITQ;
        c0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        c1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        float4 f0[NQ4], f1[NQ4];
#pragma unroll
        for (int f = 0; f < NQ4; ++f) {
            f0[f] = *reinterpret_cast<const float4 *>(b0 + 4 * f);
            f1[f] = *reinterpret_cast<const float4 *>(b1 + 4 * f);
        }
#pragma unroll
        for (int f = 0; f < NQ4; ++f)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(a[f], e), f4e(f0[f], e), c0, 0, 0, 0);
                if (two) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(a[f], e), f4e(f1[f], e), c1, 0, 0, 0);
            }
    };
    const int il_base = rt * 16 + 4 * kq;                           // C layout: column = lane & 15, row = 4*(lane>>4) + r
    const int Tpad4 = (T + 3) & ~3;
    const int w_lo = i0 + rt * 16, w_hi = w_lo + 15;                // this wave's query rows need p in [wpmin, wpmax]
    const int wpmin = T - 1 - w_hi, wpmax = 2 * T - 2 - w_lo;
    f32x4 acc[NDV];
#pragma unroll
    for (int m = 0; m < NDV; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float *sa = S + kq * SPLANE + (rt * 16 + l15) * PITS;    // AV A operand: row il = rt*16 + l15, k = 4s + kq at float s

    for (int job = 0; job < njobs; ++job) {
        if (job + 1 < njobs) issue(job + 1);
        if (job < nK) {
            // ---- content scores (q+u) K^T -> S ------------------------------------------------------------------------
            const int c0r = job * CH;
            const int nt = ((T - c0r < CH ? T - c0r : CH) + 15) / 16;   // column tiles of this chunk that hold keys
            for (int t = cp; t < nt; t += 4) {
                const bool two = t + 2 < nt;
                f32x4 a0, a1;
                tile_pair(qu, t, two ? t + 2 : t, two, a0, a1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int il = il_base + r;
                    const int j0 = c0r + t * 16 + l15, j1 = j0 + 32;
                    if (j0 < Tpad4) S[sidx(il, j0)] = j0 < T ? a0[r] : 0.0f;     // columns T..Tpad4-1: zero pad of the AV chain
                    if (two && j1 < Tpad4) S[sidx(il, j1)] = j1 < T ? a1[r] : 0.0f;
                }
            }
        } else if (job < nK + nP) {
            // ---- position scores (q+v) P^T, shifted, combined and scaled ----------------------------------------------
            const int p0 = p_first + (job - nK) * CH;
            int nt = (pmax - p0) / 16 + 1;
            nt = nt < CH / 16 ? nt : CH / 16;
            for (int t = cp; t < nt; t += 4) {
                const bool two = t + 2 < nt;
                const int pt0 = p0 + t * 16, pt1 = pt0 + 32;
                const bool need0 = pt0 + 15 >= wpmin && pt0 <= wpmax;           // wave-uniform
                const bool need1 = two && pt1 + 15 >= wpmin && pt1 <= wpmax;
                if (!need0 && !need1) continue;
                f32x4 a0, a1;
                tile_pair(qv, t, two ? t + 2 : t, two, a0, a1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int il = il_base + r, i = i0 + il;
                    const int pa = pt0 + l15, ja = pa - (T - 1) + i;
                    if (need0 && i < T && pa < P && ja >= 0 && ja < T) {
                        const int a = sidx(il, ja);
                        S[a] = (S[a] + a0[r]) * scale;                          // (content + pos) * scale, src/encoder.cpp:157-160
                    }
                    const int pb_ = pt1 + l15, jb = pb_ - (T - 1) + i;
                    if (need1 && i < T && pb_ < P && jb >= 0 && jb < T) {
                        const int a = sidx(il, jb);
                        S[a] = (S[a] + a1[r]) * scale;
                    }
                }
            }
        } else {
            // ---- ctx += softmax(S) V over this chunk (k = key index, natural order; NDV independent column tiles) ------
            const int c0r = (job - nK - nP) * CH;
            const int s_end = ((T - c0r < CH ? T - c0r : CH) + 3) / 4;          // k-steps (zero rows pad the last one)
            const float *vrow = KP + kq * VPIT + cp * 16 + l15;                 // B: V[4s + kq - c0r][dv tile (cp + 2m)]
            for (int s4 = 0; s4 < s_end; s4 += 4) {
                const float4 a = *reinterpret_cast<const float4 *>(sa + c0r / 4 + s4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (s4 + e < s_end) {
#pragma unroll
                        for (int m = 0; m < NDV; ++m)
                            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(a, e), vrow[(s4 + e) * 4 * VPIT + m * 32], acc[m], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
        if (job + 1 < njobs) commit(job + 1);
        if (job == nK + nP - 1) {
            // ---- softmax, one wavefront per row (scores complete; V chunk 0 is being committed) ----------------------------
            for (int il = wave; il < rows; il += 4) {
                float m = -__builtin_huge_valf();
                for (int j = lane; j < T; j += 64) m = fmaxf(m, S[sidx(il, j)]);
                m = wave_max64(m);
                float p = 0.0f;
                for (int j = lane; j < T; j += 64) {
                    const int a = sidx(il, j);
                    const float e = dexpf(S[a] - m);
                    S[a] = e;
                    p = p + e;
                }
                const float sum = wave_sum64(p);
                for (int j = lane; j < T; j += 64) {
                    const int a = sidx(il, j);
                    S[a] = S[a] / sum;
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < NDV; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int il = il_base + r;
            if (il < rows) ctx[((int64_t)b * T + i0 + il) * d + h * HD + (cp + 2 * m) * 16 + l15] = acc[m][r];
        }
}

template <int HD, int CH>
static void launch_att(const float *qkv, int B, int T, int d, int n_heads, const float *pos, const float *bias_u, const float *bias_v,
                       float *ctx, hipStream_t s) {
    const float scale = 1.0f / sqrtf((float)HD);                  // src/encoder.cpp:126
    int pits = (T + 3) / 4;                                        // floats per score-plane row, padded so that pits/4 is odd
    pits = (pits + 3) & ~3;
    if (((pits / 4) & 1) == 0) pits += 4;
    constexpr int PITQ = HD / 4 + 4;
    int s_floats = 4 * RB * pits;
    if (s_floats < 4 * RB * PITQ) s_floats = 4 * RB * PITQ;        // the Q tile is staged in the score block's space
    const int kp_floats = CH * (HD + 16);                          // = 4 * CH * PITQ
    const size_t lds = (size_t)(s_floats + kp_floats) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&relpos_attention_kernel<HD, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    const int n_rb = (T + RB - 1) / RB, n_bh = B * n_heads;
    dim3 grid(((n_bh + 7) / 8) * 8 * n_rb);                        // 8 XCD lanes x ceil(n_bh/8) pairs x n_rb row blocks
    hipLaunchKernelGGL((relpos_attention_kernel<HD, CH>), grid, dim3(256), lds, s, qkv, 3 * d, d, T, pos, bias_u, bias_v, scale, ctx, pits, s_floats,
                       n_rb, n_bh);
}

void launch_relpos_attention(const float *qkv, int B, int T, int d, int n_heads, const float *pos, const float *bias_u,
                             const float *bias_v, float *ctx, hipStream_t s) {
    const int hd = d / n_heads;
    // chunk rows: hd = 64 with T <= 129 (10 s clips: T = 126) holds K, the whole T+31-row P band and V in ONE chunk each
    // (160 rows, 70 KB with the score block -> 2 workgroups per CU); longer sequences stream 128-row chunks.
    if (hd == 64 && T <= 129) launch_att<64, 160>(qkv, B, T, d, n_heads, pos, bias_u, bias_v, ctx, s);
    else if (hd == 64) launch_att<64, 128>(qkv, B, T, d, n_heads, pos, bias_u, bias_v, ctx, s);
    else if (hd == 128) launch_att<128, 64>(qkv, B, T, d, n_heads, pos, bias_u, bias_v, ctx, s);
    else if (hd == 32) launch_att<32, 128>(qkv, B, T, d, n_heads, pos, bias_u, bias_v, ctx, s);
    else if (hd == 96) launch_att<96, 64>(qkv, B, T, d, n_heads, pos, bias_u, bias_v, ctx, s);
}

}  // namespace pk
