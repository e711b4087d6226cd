// parakeet.cpp_amd/csrc/kernels/attention.hip -- relative-position multi-head attention core
// (reference ConformerAttention::rel_position_attention, src/encoder.cpp:135-171, with rel_shift
// :85-109 applied in closed form):
//     S[i][j] = ( (q_i + u_h) . k_j  +  (q_i + v_h) . P_h[j - i + T - 1] ) / sqrt(hd)
//     ctx_i   = softmax_j(S[i][:]) V
// One workgroup per (utterance, head, block of 32 query rows), 4 wavefronts.  All three contractions run on the
// fp32 MFMA v_mfma_f32_16x16x4_f32 (natural-k fma chains = the oracle's order); 16x16 tiles give every wave the same
// number of tiles in each phase.
//   * Q, K and the projected position table P arrive in the "sigma" column layout (the producing GEMMs write it: inside
//     every block of 16 features the 4x4 index matrix is transposed), so the float4 a lane loads holds exactly its
//     k = 4s+kq operands of four consecutive MFMA steps: the A / B fragments of QK^T and QP^T come STRAIGHT from L2 with
//     16-byte loads -- no LDS staging, no barriers, each wave streams its own tiles with the next pair's loads in flight.
//   * The [32][T] score block lives in LDS only.  The [B][H][T][2T-1] position-score tensor the reference materialises is
//     never formed: each 16x16 position tile is added straight into its shifted column j = p - (T-1) + i.  The block is
//     "k-planar" over the key index (element (i, j) at plane j&3, i*pitch + (j>>2)) because it is the A operand of
//     softmax(S) V: one ds_read_b128 feeds four MFMA steps.
//   * V streams through LDS in chunks of VCH rows (coalesced 16-byte loads, natural rows, pitch = 16 mod 32 banks).
// Softmax is one wavefront per row with the canonical max / sum64 butterflies.  3 barriers per workgroup (+2 per extra V chunk).
// (Round 2 also tried V as a straight-from-L2 B operand -- one dword per lane per MFMA step, a register block of 8 steps ahead: bit-equal,
// no V copy in LDS and no chunk barriers, but the gathers are slower than the LDS copy: AV phase 10 k -> 23 k clocks per wave at T = 126,
// 52 k -> 112 k at T = 376 / hd = 128.  Not kept.)
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
static constexpr int RB = 32;   // query rows per workgroup
#ifndef ATT_OCC
#define ATT_OCC 4                 // workgroups per CU the hd <= 64 kernel is compiled for (register budget 168 / 128 VGPRs for 3 / 4)
#endif

// Phase stamps for tools/ubench/attn_bench.cpp (-DATT_TRACE): shader clock of lane 0 of every wave at the phase boundaries.  Production: nothing.
#ifdef ATT_TRACE
__device__ long long *att_trace;    // [workgroup][wave][8]
#define ATT_STAMP(i) do { if (att_trace && (threadIdx.x & 63) == 0) att_trace[((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] = clock64(); } while (0)
#else
#define ATT_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ float f4e(const float4 &v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// SG (long sequences, > ~85 s of audio): the score block of a workgroup lives in a global scratch area instead of LDS -- the same
// code, indexing and barriers (a workgroup's own global writes are visible to it after __syncthreads()), so the same bits; slower,
// but the length is then bounded by HBM, not by the 160 KB of LDS.
// RAG: the ragged-batch form (per-unit utterance extents, kernels.hpp: SeqRag) is its own instantiation -- the uniform kernel sits exactly at
// its 128-VGPR budget (four workgroups per CU) and must not carry a single extra live value (round 4: the shared form spilled 11 dwords per
// lane instead of 4 and lost 2.5 %).
// OCC: workgroups per CU the instantiation is register-budgeted for (128 / 168 / 256 VGPRs for 4 / 3 / 2).  The launcher picks the largest the
// LDS footprint of the sequence length allows anyway: a 30 s utterance (T = 376: 71 KB of score planes + V chunk, two workgroups per CU)
// gains nothing from the spills of the 128-VGPR build.
template <int HD, int VCH, bool SG = false, bool RAG = false, int OCC = (HD <= 64 ? ATT_OCC : 2)>
__global__ __launch_bounds__(256, OCC) void relpos_attention_kernel(const float *__restrict__ qkv, int ldq, int d, int T,
                                                               const float *__restrict__ pos /*[2T-1][d], sigma columns*/,
                                                               const float *__restrict__ bias_u, const float *__restrict__ bias_v,
                                                               float scale, float *__restrict__ ctx, int PITS, int n_rb, int n_bh,
                                                               float *__restrict__ s_scratch, int ctx_bf16, int pos_row0, SeqRag rg) {
    __builtin_amdgcn_s_setprio(3);        // ahead of the overlapped decode-loop waves in the SIMD's arbitration (gemm_pipe.hpp)
    constexpr int KQ = HD / 4;
    constexpr int NQ4 = HD / 16;          // float4 fragments per lane for K = HD (one per block of 16 features)
    constexpr int VPIT = HD + 16;         // V rows, natural layout (pitch = 16 mod 32 banks)
    constexpr int NDV = HD / 32;          // 16-wide ctx column tiles per wave (HD/16 tiles over 2 column parities)
    constexpr int NLV = VCH * KQ / 256;   // float4 loads per thread for one V chunk
    static_assert((VCH * KQ) % 256 == 0 && VCH % 4 == 0, "V chunk must split evenly over 256 threads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int SPLANE = RB * PITS + 8;     // +8: the four planes start 8 banks apart (conflict-free row sweeps)
    float *S = SG ? s_scratch + (size_t)blockIdx.x * 4 * SPLANE : smem;     // [4][SPLANE] score planes
    float *VS = SG ? smem : smem + 4 * SPLANE;                               // [VCH][VPIT] V chunk
    const int H = d / HD;
    // Block b runs on XCD b % 8 (observed dispatch rule).  The row blocks of one (utterance, head) share K, V and the P band:
    // give them consecutive slots on ONE XCD so the second..last read those rows from that XCD's L2 instead of HBM.
    int h, i0;
    int64_t row0;                                                   // first row of this utterance in the (packed) row axis of qkv / ctx
    if constexpr (RAG) {
        // ragged batch (kernels.hpp: SeqRag): XCD x takes head x (+ 8, ...) of EVERY utterance, the row blocks of one utterance in consecutive
        // slots -- the same L2 sharing; this utterance's own length, and its window of the position table built for rg.pos_T frames
        const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
        h = (k / rg.units.count) * 8 + xcd;
        if (h >= H) return;
        const RagUnit un = rg.units.u[k % rg.units.count];
        i0 = un.r0;
        T = rg.T[un.b];
        row0 = rg.T_off[un.b];
        pos_row0 = rg.pos_T - T;
    } else {
        const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
        const int rbk = k % n_rb, bh = (k / n_rb) * 8 + xcd;
        if (bh >= n_bh) return;                                      // padding slot of the grid (whole workgroup, before any barrier)
        h = bh % H;
        i0 = rbk * RB;
        row0 = (int64_t)(bh / H) * T;
    }
    const int rows = (T - i0) < RB ? (T - i0) : RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int rt = wave & 1, cp = wave >> 1;                       // this wave's 16-row tile and tile parity
    const int P = 2 * T - 1;
    const float *qb = qkv + row0 * ldq + h * HD;                    // q rows of this (b,h) (sigma columns)
    const float *kb = qb + d, *vb = qb + 2 * d;                     // k (sigma columns), v (natural)
    const float *pb = pos + (int64_t)pos_row0 * d + h * HD;         // row p of THIS length's table (engine.cpp: ensure_pos_tables)
    auto sidx = [&](int il, int j) { return (j & 3) * SPLANE + il * PITS + (j >> 2); };

    // V chunk: cooperative coalesced loads -> registers -> LDS
    float4 vf[NLV];
    auto v_issue = [&](int row0) {
#pragma unroll
        for (int i = 0; i < NLV; ++i) {
            const int e = tid + 256 * i, gr = row0 + e / KQ;
            vf[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (gr < T) vf[i] = *reinterpret_cast<const float4 *>(vb + (int64_t)gr * ldq + 4 * (e % KQ));
        }
    };
    auto v_commit = [&]() {
#pragma unroll
        for (int i = 0; i < NLV; ++i) {
            const int e = tid + 256 * i;
            lds_store16(VS + (e / KQ) * VPIT + 4 * (e % KQ), vf[i]);
        }
    };
    // one 16-row operand tile straight from global: lane (row l15, quarter kq) takes float4 #kq of every 16-feature block
    auto load_tile = [&](const float *base, int64_t ld, int row0, int limit, float4 (&f)[NQ4]) {
#ifdef ATT_LINES_EXPERIMENT     // TIMING PROXY ONLY (wrong operands): the same number of 16-byte loads, but every instruction reads 8 rows x one whole 128-byte line
        const int r = row0 + (lane >> 3);
        const bool ok = r >= 0 && r + 8 < limit;
        const float *p = base + (int64_t)(ok ? r : 0) * ld + 4 * (lane & 7);
#pragma unroll
        for (int q = 0; q < NQ4; ++q) f[q] = ok ? *reinterpret_cast<const float4 *>(p + (int64_t)(8 * (q & 1)) * ld + 32 * (q >> 1)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#else
        const int r = row0 + l15;
        const bool ok = r >= 0 && r < limit;
        const float *p = base + (int64_t)(ok ? r : 0) * ld + 4 * kq;
#pragma unroll
        for (int q = 0; q < NQ4; ++q) f[q] = ok ? *reinterpret_cast<const float4 *>(p + 16 * q) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#endif
    };
    // two independent 16x16 accumulator chains over K = HD (natural k: step 4q+e consumes k = 16q + 4e + kq)
    auto mma_pair = [&](const float4 (&a)[NQ4], const float4 (&b0)[NQ4], const float4 (&b1)[NQ4], f32x4 &c0, f32x4 &c1) {
        c0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        c1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < NQ4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(a[q], e), f4e(b0[q], e), c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f4e(a[q], e), f4e(b1[q], e), c1, 0, 0, 0);
            }
    };

    // ---- phase 0: the Q fragments of this wave's 16 rows.  Only ONE biased copy is live at a time -- (q+u) for the content scores, then
    //      q is fetched again (L2) and (q+v) formed for the position scores -- and the first V chunk is requested after the score phases:
    //      the register budget of four workgroups per CU (128 VGPRs) has no room for both copies plus the V prefetch beside the four
    //      operand tiles of the score loops (round 2: 168 -> 134 VGPRs, 86.9 -> 82.3 us per layer at T = 126).
    float4 qx[NQ4];                                                 // (q+u), later (q+v); element e of fragment f <-> k = 16f + 4e + kq
    auto load_q_biased = [&](const float *bias) {                   // as the reference forms them (src/encoder.cpp:141-142)
        load_tile(qb, ldq, i0 + rt * 16, T, qx);
        if (pos) {
            const float *br = bias + h * HD + kq;
#pragma unroll
            for (int f = 0; f < NQ4; ++f)
                qx[f] = make_float4(qx[f].x + br[16 * f], qx[f].y + br[16 * f + 4], qx[f].z + br[16 * f + 8], qx[f].w + br[16 * f + 12]);
        }
    };
    ATT_STAMP(0);
    load_q_biased(bias_u);
    const int il_base = rt * 16 + 4 * kq;                           // C layout: column = lane & 15, row = 4*(lane>>4) + r
    const int Tpad4 = (T + 3) & ~3;
    ATT_STAMP(1);                                                   // Q load + bias
    float4 bA0[NQ4], bA1[NQ4], bB0[NQ4], bB1[NQ4];                  // two operand-tile pairs: one computing, one in flight

    // ---- phase 1: content scores (q+u) K^T -> S; this wave's column tiles are t = cp, cp+2, ... (pairs, one pair ahead) ----
    {
        const int nct = (T + 15) / 16;
        auto store = [&](int t, const f32x4 &a0, const f32x4 &a1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int il = il_base + r;
                const int j0 = t * 16 + l15, j1 = j0 + 32;
                // columns T..Tpad4-1: zero pad of the AV chain; without a position term the scale is applied here
                if (j0 < Tpad4) S[sidx(il, j0)] = j0 < T ? (pos ? a0[r] : a0[r] * scale) : 0.0f;
                if (t + 2 < nct && j1 < Tpad4) S[sidx(il, j1)] = j1 < T ? (pos ? a1[r] : a1[r] * scale) : 0.0f;
            }
        };
        if (cp < nct) { load_tile(kb, ldq, cp * 16, T, bA0); load_tile(kb, ldq, (cp + 2) * 16, T, bA1); }
        for (int t = cp; t < nct; t += 8) {
            f32x4 a0, a1;
            if (t + 4 < nct) { load_tile(kb, ldq, (t + 4) * 16, T, bB0); load_tile(kb, ldq, (t + 6) * 16, T, bB1); }
            mma_pair(qx, bA0, bA1, a0, a1);
            store(t, a0, a1);
            if (t + 4 < nct) {
                if (t + 8 < nct) { load_tile(kb, ldq, (t + 8) * 16, T, bA0); load_tile(kb, ldq, (t + 10) * 16, T, bA1); }
                mma_pair(qx, bB0, bB1, a0, a1);
                store(t + 4, a0, a1);
            }
        }
    }
    ATT_STAMP(2);                                                   // QK^T -> S
    if (pos) load_q_biased(bias_v);
    // ---- phase 2: position scores (q+v) P^T, shifted, combined and scaled.  This wave's 16 query rows need
    //      p = j - i + T - 1 in [wpmin, wpmax] (T+15 rows): its own tile grid starts at wpmin, tiles t = cp, cp+2, ... ------
    const int w_lo = i0 + rt * 16, w_hi = (w_lo + 15) < (T - 1) ? (w_lo + 15) : (T - 1);
    const int wpmin = T - 1 - w_hi, wpmax = 2 * T - 2 - w_lo;
    const int npt = (pos && w_lo < T) ? (wpmax - wpmin) / 16 + 1 : 0;
    if (cp < npt) { load_tile(pb, d, wpmin + cp * 16, P, bA0); load_tile(pb, d, wpmin + (cp + 2) * 16, P, bA1); }   // in flight across the barrier
    __syncthreads();                                              // content scores complete
    ATT_STAMP(3);                                                   // (q+v), first P tiles requested, barrier
    {
        auto rmw = [&](int t, const f32x4 &a0, const f32x4 &a1) {
            // the eight score elements are READ together, then combined, then stored (distinct addresses: four rows x two column tiles); one by one
            // every read would wait behind the previous element's store (see the softmax sweep below)
            int ad[8];
            bool ok[8];
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int il = il_base + r, i = i0 + il;
                const int pa = wpmin + t * 16 + l15, ja = pa - (T - 1) + i;
                const int pc = pa + 32, jc = ja + 32;
                ok[r] = i < T && pa < P && ja >= 0 && ja < T;
                ok[4 + r] = t + 2 < npt && i < T && pc < P && jc >= 0 && jc < T;
                ad[r] = ok[r] ? sidx(il, ja) : 0;
                ad[4 + r] = ok[4 + r] ? sidx(il, jc) : 0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = S[ad[q]];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (ok[r]) S[ad[r]] = (v[r] + a0[r]) * scale;                   // (content + pos) * scale, src/encoder.cpp:157-160
                if (ok[4 + r]) S[ad[4 + r]] = (v[4 + r] + a1[r]) * scale;
            }
        };
        for (int t = cp; t < npt; t += 8) {
            f32x4 a0, a1;
            if (t + 4 < npt) { load_tile(pb, d, wpmin + (t + 4) * 16, P, bB0); load_tile(pb, d, wpmin + (t + 6) * 16, P, bB1); }
            mma_pair(qx, bA0, bA1, a0, a1);
            rmw(t, a0, a1);
            if (t + 4 < npt) {
                if (t + 8 < npt) { load_tile(pb, d, wpmin + (t + 8) * 16, P, bA0); load_tile(pb, d, wpmin + (t + 10) * 16, P, bA1); }
                mma_pair(qx, bB0, bB1, a0, a1);
                rmw(t + 4, a0, a1);
            }
        }
    }
    ATT_STAMP(4);                                                   // QP^T shifted rmw
    v_issue(0);                                                   // first V chunk: in flight across the softmax
    __syncthreads();
    ATT_STAMP(5);                                                   // barrier
    // ---- phase 3: softmax, one wavefront per row ----------------------------------------------------------------------------
    // Each wave owns rows wave, wave+4, ...: all NSR of them go through the three sweeps TOGETHER, so the 2 x 6 dependent
    // cross-lane butterfly stages and the exp / divide chains of different rows overlap instead of queueing up.
    // (Rows past T in the last block hold zeros: they are swept too and never stored.)
    {
        constexpr int NSR = RB / 4;
        float mx[NSR], sm[NSR];
#pragma unroll
        for (int k = 0; k < NSR; ++k) mx[k] = -__builtin_huge_valf();
        for (int j = lane; j < T; j += 64)
#pragma unroll
            for (int k = 0; k < NSR; ++k) mx[k] = fmaxf(mx[k], S[sidx(wave + 4 * k, j)]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int k = 0; k < NSR; ++k) mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
#pragma unroll
        for (int k = 0; k < NSR; ++k) sm[k] = 0.0f;
        // The NSR rows' values are READ first, then transformed, then stored: written element by element (read, exp, store, next row) the compiler
        // has to keep every read behind the previous row's store -- both are LDS floats -- and the eight exp chains ran one after the other, each
        // behind its own LDS round trip (round 4, from the ISA: profiles/r04_attn_phases.txt).  Same values, same per-row summation order.
        for (int j = lane; j < T; j += 64) {
            float e[NSR];
#pragma unroll
            for (int k = 0; k < NSR; ++k) e[k] = S[sidx(wave + 4 * k, j)] - mx[k];          // S <= row maximum
            if (ctx_bf16 == 1) {    // bf16 mode (tolerance-class, see GemmArgs::fast_act): hardware exp2 instead of the fixed polynomial
#pragma unroll
                for (int k = 0; k < NSR; ++k) e[k] = __builtin_amdgcn_exp2f(e[k] * 1.44269502162933349609375f);
            } else {
#pragma unroll
                for (int k = 0; k < NSR; ++k) e[k] = dexpf_nonpos(e[k]);
            }
#pragma unroll
            for (int k = 0; k < NSR; ++k) {
                S[sidx(wave + 4 * k, j)] = e[k];
                sm[k] = sm[k] + e[k];
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)                      // the canonical sum64 butterfly, NSR rows side by side
#pragma unroll
            for (int k = 0; k < NSR; ++k) sm[k] = sm[k] + __shfl_xor(sm[k], off, 64);
        if (ctx_bf16 == 1) {                                         // bf16 mode: one hardware reciprocal per row, a multiplication per element
#pragma unroll
            for (int k = 0; k < NSR; ++k) sm[k] = __builtin_amdgcn_rcpf(sm[k]);
            for (int j = lane; j < T; j += 64) {
                float e[NSR];
#pragma unroll
                for (int k = 0; k < NSR; ++k) e[k] = S[sidx(wave + 4 * k, j)] * sm[k];
#pragma unroll
                for (int k = 0; k < NSR; ++k) S[sidx(wave + 4 * k, j)] = e[k];
            }
        } else {
            for (int j = lane; j < T; j += 64) {
                float e[NSR];
#pragma unroll
                for (int k = 0; k < NSR; ++k) e[k] = S[sidx(wave + 4 * k, j)];
#pragma unroll
                for (int k = 0; k < NSR; ++k) e[k] = e[k] / sm[k];
#pragma unroll
                for (int k = 0; k < NSR; ++k) S[sidx(wave + 4 * k, j)] = e[k];
            }
        }
    }
    ATT_STAMP(6);                                                   // softmax
    v_commit();
    lds_store_fence();
    __syncthreads();
    // ---- phase 4: ctx = softmax(S) V  (k = key index, natural order; NDV independent 16-column tiles per wave) ---------------
    f32x4 acc[NDV];
#pragma unroll
    for (int m = 0; m < NDV; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const float *sa = S + kq * SPLANE + (rt * 16 + l15) * PITS;    // A: row il = rt*16 + l15, k = 4s + kq at float s
    const float *vrow = VS + kq * VPIT + cp * 16 + l15;             // B: V[4s + kq - c0r][dv tile (cp + 2m)]
    for (int c0r = 0; c0r < T; c0r += VCH) {
        const bool more = c0r + VCH < T;
        if (more) v_issue(c0r + VCH);
        const int s_end = ((T - c0r < VCH ? T - c0r : VCH) + 3) / 4;            // k-steps in this chunk (zero rows pad the last one)
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
        if (more) {
            __syncthreads();
            v_commit();
            lds_store_fence();
            __syncthreads();
        }
    }
#pragma unroll
    for (int m = 0; m < NDV; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int il = il_base + r;
            if (il < rows) {
                const int64_t orow = (row0 + i0 + il) * d;
                const int col = h * HD + (cp + 2 * m) * 16 + l15;
                if (ctx_bf16 == 1) reinterpret_cast<__bf16 *>(ctx)[orow + col] = (__bf16)acc[m][r];      // bf16 mode: out_proj's operand, rounded here (RNE)
                else if (ctx_bf16 == 2) ctx[orow + ((col & ~15) | ((col & 3) << 2) | ((col >> 2) & 3))] = acc[m][r];   // fp32, sigma K layout (GemmArgs::a_sigma)
                else ctx[orow + col] = acc[m][r];
            }
        }
    ATT_STAMP(7);                                                   // V commit, barrier, AV, store
}

template <int HD, int VCH, int OCC = (HD <= 64 ? ATT_OCC : 2)>
static void launch_att(const float *qkv, int B, int T, int d, int n_heads, const float *pos, const float *bias_u, const float *bias_v,
                       float *ctx, float scale_arg, hipStream_t s, float *scratch, int ctx_bf16, int pos_row0, const SeqRag &rag) {
    const float scale = scale_arg > 0.0f ? scale_arg : 1.0f / sqrtf((float)HD);   // src/encoder.cpp:126
    if (rag.units.u) T = rag.T_max;                                // the score block is laid out for the longest utterance of the batch
    int pits = (T + 3) / 4;                                        // floats per score-plane row, padded so that pits/4 is odd
    pits = (pits + 3) & ~3;
    if (((pits / 4) & 1) == 0) pits += 4;
    const int n_rb = (T + RB - 1) / RB, n_bh = B * n_heads;
    dim3 grid(((n_bh + 7) / 8) * 8 * n_rb);                        // 8 XCD lanes x ceil(n_bh/8) pairs x n_rb row blocks
    if (rag.units.u) grid = dim3((unsigned)(((n_heads + 7) / 8) * 8 * (int64_t)rag.units.count));
    if (scratch) {
        {
            const size_t lds = (size_t)VCH * (HD + 16) * sizeof(float);
            if (rag.units.u)
                hipLaunchKernelGGL((relpos_attention_kernel<HD, VCH, true, true, OCC>), grid, dim3(256), lds, s, qkv, 3 * d, d, T, pos, bias_u, bias_v, scale, ctx, pits,
                                   n_rb, n_bh, scratch, ctx_bf16, pos_row0, rag);
            else
                hipLaunchKernelGGL((relpos_attention_kernel<HD, VCH, true, false, OCC>), grid, dim3(256), lds, s, qkv, 3 * d, d, T, pos, bias_u, bias_v, scale, ctx, pits,
                                   n_rb, n_bh, scratch, ctx_bf16, pos_row0, rag);
        }
        return;
    }
    const size_t lds = (size_t)(4 * (RB * pits + 8) + VCH * (HD + 16)) * sizeof(float);
    if (rag.units.u) {
        static DynLdsSlots slots_r;
        ensure_dyn_lds(slots_r, reinterpret_cast<const void *>(&relpos_attention_kernel<HD, VCH, false, true, OCC>), lds);
        hipLaunchKernelGGL((relpos_attention_kernel<HD, VCH, false, true, OCC>), grid, dim3(256), lds, s, qkv, 3 * d, d, T, pos, bias_u, bias_v, scale, ctx, pits, n_rb, n_bh,
                           (float *)nullptr, ctx_bf16, pos_row0, rag);
        return;
    }
    static DynLdsSlots slots;
    ensure_dyn_lds(slots, reinterpret_cast<const void *>(&relpos_attention_kernel<HD, VCH, false, false, OCC>), lds);
    hipLaunchKernelGGL((relpos_attention_kernel<HD, VCH, false, false, OCC>), grid, dim3(256), lds, s, qkv, 3 * d, d, T, pos, bias_u, bias_v, scale, ctx, pits, n_rb, n_bh,
                       (float *)nullptr, ctx_bf16, pos_row0, rag);
}

// LDS bytes one workgroup needs for T frames: the [32][T] score block (four k-planes) + one V chunk.  0: unsupported head size.
size_t relpos_attention_lds_bytes(int T, int hd) {
    if (hd != 32 && hd != 64 && hd != 96 && hd != 128) return 0;
    int pits = (T + 3) / 4;
    pits = (pits + 3) & ~3;
    if (((pits / 4) & 1) == 0) pits += 4;
    const int vch = hd == 128 ? 32 : 64;
    return (size_t)(4 * (RB * pits + 8) + vch * (hd + 16)) * sizeof(float);
}
// longest sequence whose score block fits the 160 KB of LDS of a CU (hd 64: 1064 frames = 85 s of audio; hd 128: 1104)
// bytes of global scratch the long-sequence variant needs (0: unsupported head size)
size_t relpos_attention_scratch_bytes(int B, int T, int n_heads, int hd) {
    if (hd != 32 && hd != 64 && hd != 96 && hd != 128) return 0;
    int pits = (T + 3) / 4;
    pits = (pits + 3) & ~3;
    if (((pits / 4) & 1) == 0) pits += 4;
    const size_t n_rb = (T + RB - 1) / RB, n_bh = (size_t)B * n_heads;
    return ((n_bh + 7) / 8) * 8 * n_rb * 4 * (size_t)(RB * pits + 8) * sizeof(float);
}
size_t relpos_attention_scratch_bytes_units(int64_t n_units, int T_max, int n_heads, int hd) {
    if (hd != 32 && hd != 64 && hd != 96 && hd != 128) return 0;
    int pits = (T_max + 3) / 4;
    pits = (pits + 3) & ~3;
    if (((pits / 4) & 1) == 0) pits += 4;
    return (size_t)((n_heads + 7) / 8) * 8 * (size_t)n_units * 4 * (size_t)(RB * pits + 8) * sizeof(float);
}
int relpos_attention_max_frames(int hd) {
    int T = 0;
    while (relpos_attention_lds_bytes(T + 8, hd) && relpos_attention_lds_bytes(T + 8, hd) <= 160 * 1024) T += 8;
    return T;
}

void launch_relpos_attention(const float *qkv, int B, int T, int d, int n_heads, const float *pos, const float *bias_u,
                             const float *bias_v, float *ctx, hipStream_t s, float scale, float *scratch, int ctx_bf16, int pos_row0, const SeqRag &rag) {
    const int hd = d / n_heads;
    // V chunk rows: 64 keeps the footprint at ~39 KB for 10 s clips (4 workgroups per CU at hd = 64).  Between ~10.5 and ~16 s (T = 132 .. 200:
    // the longest clip of a mixed batch, 15 s uniform batches) the 64-row chunk costs the fourth resident workgroup; a 32-row chunk (two more
    // barriers per extra chunk, the same chains: same bits) keeps it
    if (hd == 64) {
        const int t_lds = rag.units.u ? rag.T_max : T;
        auto occ = [&](int vch) {
            int pits = (t_lds + 3) / 4;
            pits = (pits + 3) & ~3;
            if (((pits / 4) & 1) == 0) pits += 4;
            const size_t lds = (size_t)(4 * (RB * pits + 8) + vch * (64 + 16)) * sizeof(float);
            const int n = (int)((size_t)160 * 1024 / lds);
            return n < ATT_OCC ? n : ATT_OCC;
        };
        // (measured on a 5-15 s mixed batch, T_max = 185: 32-row chunks at four workgroups per CU 1.69 ms, 64-row chunks at three without spills 1.77,
        //  64-row chunks at three on the 128-VGPR build 1.85)
        if (!scratch && occ(32) > occ(64)) launch_att<64, 32>(qkv, B, T, d, n_heads, pos, bias_u, bias_v, ctx, scale, s, scratch, ctx_bf16, pos_row0, rag);
        // LDS allows at most three workgroups per CU whatever the chunk (T > 200): the register budget of three (the kernel needs 134-140 VGPRs: no spills)
        else if (!scratch && occ(64) <= 3) launch_att<64, 64, 3>(qkv, B, T, d, n_heads, pos, bias_u, bias_v, ctx, scale, s, scratch, ctx_bf16, pos_row0, rag);
        else launch_att<64, 64>(qkv, B, T, d, n_heads, pos, bias_u, bias_v, ctx, scale, s, scratch, ctx_bf16, pos_row0, rag);
    } else if (hd == 128) launch_att<128, 32>(qkv, B, T, d, n_heads, pos, bias_u, bias_v, ctx, scale, s, scratch, ctx_bf16, pos_row0, rag);
    else if (hd == 32) launch_att<32, 64>(qkv, B, T, d, n_heads, pos, bias_u, bias_v, ctx, scale, s, scratch, ctx_bf16, pos_row0, rag);
    else if (hd == 96) launch_att<96, 64>(qkv, B, T, d, n_heads, pos, bias_u, bias_v, ctx, scale, s, scratch, ctx_bf16, pos_row0, rag);
}

}  // namespace pk
