// parakeet.cpp_amd/csrc/kernels/decode_dev.hpp -- device code of the TDT / RNNT decode step shared by the per-phase kernels
// (decode_gemv.hip, decode.hip) and the persistent single-launch kernel (decode_persist.hip).
#pragma once
#include "../pk_devmath.h"
#include "kernels.hpp"
#include <type_traits>

// Phase stamps for tools/ubench/decide_trace.cpp (-DDEC_TRACE): shader clock of thread 0 at the phase boundaries of tdt_decide_one, kept in registers and written
// behind the last one (a walked decision overwrites the stamps of the decision before it; [15] counts decisions).  Production: nothing.
#ifdef DEC_TRACE
__device__ long long *dec_trace;    // [16]
#define DEC_STAMP(i) do { dec_st[i] = clock64(); } while (0)
#define DEC_STAMPS_DECL long long dec_st[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define DEC_STAMPS_FLUSH() do { if (dec_trace && threadIdx.x == 0) { for (int i_ = 0; i_ < 16; ++i_) dec_trace[i_] = dec_st[i_]; } } while (0)
#else
#define DEC_STAMP(i) do { } while (0)
#define DEC_STAMPS_DECL do { } while (0)
#define DEC_STAMPS_FLUSH() do { } while (0)
#endif

namespace pk {

// accesses to data exchanged between workgroups inside one launch: system scope (sc0 sc1) relaxed atomics, which the compiler
// neither caches nor serialises (a volatile access would be followed by s_waitcnt vmcnt(0)); plain accesses otherwise
template <bool COH> __device__ __forceinline__ float dd_ldf(const float *p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else return *p;
}
template <bool COH> __device__ __forceinline__ int dd_ldi(const int *p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else return *p;
}
template <bool COH> __device__ __forceinline__ float4 dd_ld4(const float4 *p) {
    if constexpr (COH) {
        const float *f = reinterpret_cast<const float *>(p);
        return make_float4(dd_ldf<true>(f), dd_ldf<true>(f + 1), dd_ldf<true>(f + 2), dd_ldf<true>(f + 3));
    } else {
        return *p;
    }
}
template <bool COH> __device__ __forceinline__ void dd_stf(float *p, float v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else *p = v;
}
template <bool COH> __device__ __forceinline__ void dd_sti(int *p, int v) {
    if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else *p = v;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16 bytes of the weight stream: plain, or with the non-temporal (streaming) hint
template <bool NT> __device__ __forceinline__ float4 dd_ldw(const float4 *p) {
    if constexpr (NT) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    } else {
        return *p;
    }
}

__device__ __forceinline__ int sigma16(int k) { return (k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3); }

// Compact list of the utterances whose need flag is set, ascending (TdtState::need): lst[0 .. count).  One 256-thread workgroup, B <= kMaxListRows;
// every thread of the workgroup must call it (two barriers).
template <bool COH>
__device__ __forceinline__ int dd_build_rowlist(const int *need, int B, int *lst, int *wtot /* [4] */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int F = (B + 255) >> 8;                                 // consecutive flags per thread (<= 8)
    const int b0 = tid * F;
    int fl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int b = b0 + i;
        fl[i] = (i < F && b < B) ? dd_ldi<COH>(need + b) : 0;
    }
    unsigned bits = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) bits |= (fl[i] != 0 ? 1u : 0u) << i;
    const int c = __builtin_popcount(bits);
    int x = c;                                                    // inclusive scan over the wave
    if (F == 1) {                                                 // one flag per thread (B <= 256): a ballot and a population count, no lane exchange
        const unsigned long long set = __ballot(c != 0);
        x = __popcll(set & ((1ull << lane) - 1ull)) + c;
    } else {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
    }
    if (lane == 63) wtot[wave] = x;
    __syncthreads();
    int base = x - c;
    for (int w = 0; w < wave; ++w) base += wtot[w];
    const int cnt = wtot[0] + wtot[1] + wtot[2] + wtot[3];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if ((bits >> i) & 1u) lst[base++] = b0 + i;
    __syncthreads();
    return cnt;
}

// NCH: compile-time number of 64-wide K chunks (10 for K = 640: fully unrolled, counted vmcnt waits keep the next
// chunk's loads in flight under the MFMA chain); 0 = runtime trip count (any K % 64 == 0).
// COH = the operands other workgroups of the SAME kernel produced (X, c, the token / frame words, gi of the upper LSTM layers) are
// read, and the outputs written, with system-scope accesses (sc0 sc1: no cache between the workgroups) -- the persistent decode
// kernel (decode_persist.hip) runs every phase of a step inside one launch.
// NTW: the weight stream is loaded non-temporally (streaming hint: the decode weights of the large heads -- 42 MB per symbol step for tdt-600m --
// pass through each XCD's 4 MB L2 once per step and otherwise evict the operand tiles of the encoder GEMMs running beside the loop).
// WF > 1 (SK_ACT only): the frame-window form of the joint activation (TdtState::F) -- utterance b's rows b * F + f of z take relu(enc_proj[t_b + f] + pp), f < a.F <= WF.
// PRED (a.need set, B <= 16: the whole batch is one row tile): no compacted row list -- the need flags are requested FIRST, together with every other operand, the
// launch leaves after the first operand chunk is under way if no flag is set, and the epilogue stores only the flagged rows.  Building the list first put a memory
// round trip (flags) in front of the launch's own loads: ~1 us of a ~5 us launch, twice per symbol step.  A row's chain does not depend on its neighbours: same bits.
template <int EPI, int NCH, bool COH, bool NTW = false, int WF = 1, bool PRED = false>
__device__ __forceinline__ void skinny_tile(const SkinnyArgs &a, int nt, int mgroup, float (*tile)[16][17], const int *rows = nullptr, int n_rows = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int m0 = (mgroup * 4 + wave) * 16;
    // rows != null: the launch covers the n_rows utterances rows[0 ..] (prediction-net caching, TdtState::need); row index i of the tile space
    // is utterance rows[i]
    const int NB = rows ? n_rows : a.B;
    auto real = [&](int i) { return rows ? rows[i] : i; };
    if constexpr (PRED) {
        if (wave) return;                                        // one row tile (1 <= B <= 16): wave 0 has it -- decided without an argument in front of the argument loads
    } else {
        if (m0 >= NB) return;                                    // whole wave out of range (uniform)
    }
    int wrow;
    if (EPI == SK_CELL) wrow = (col >> 2) * a.Hp + 4 * nt + (col & 3);   // tile columns = (gate, unit): rows g*Hp + j
    else { wrow = 16 * nt + col; wrow = wrow < a.N ? wrow : a.N - 1; }
    int xrow = m0 + col;
    xrow = real(xrow < NB ? xrow : NB - 1);
    const float4 *xp = reinterpret_cast<const float4 *>(a.X + (int64_t)xrow * a.K) + kq;
    const float4 *wp = reinterpret_cast<const float4 *>(a.W + (int64_t)wrow * a.K) + kq;
    // Epilogue operands are fetched FIRST (token -> g1 row, c, enc_proj[t_b], bias): in this latency-bound loop every
    // dependent round trip to L2 / HBM that can hide under the 160-MFMA chain is ~1-2 us saved per launch.
    float e_gi[4] = {0.0f, 0.0f, 0.0f, 0.0f}, e_c = 0.0f;          // SK_CELL: lane -> (utterance lane>>2, unit lane&3)
    float e_ep[4] = {0.0f, 0.0f, 0.0f, 0.0f}, e_bias = 0.0f;       // SK_ACT / SK_BIAS: lane -> column `col`, utterances 4*kq+r
    float e_epw[4][WF > 1 ? WF - 1 : 1];                           // SK_ACT window: enc_proj of the frames t + 1 .. t + F - 1
    int rb_cell = 0, rb_out[4] = {0, 0, 0, 0};                   // utterances of this lane's epilogue rows
    int nd_cell = 1, nd_out[4] = {1, 1, 1, 1};                   // PRED: their need flags
    bool pred_skip = false;
    if (EPI == SK_CELL) {
        const int bi = m0 + (lane >> 2), j = 4 * nt + (lane & 3);
        const int b = bi < NB ? real(bi) : 0;
        rb_cell = b;
        if constexpr (PRED) nd_cell = bi < NB ? dd_ldi<COH>(a.need + b) : 0;
        if (bi < NB && !a.W2) {
            const float *gir = a.gi + (int64_t)(a.gi_row ? dd_ldi<COH>(a.gi_row + b) : b) * a.gi_ld;
#pragma unroll
            for (int g = 0; g < 4; ++g) e_gi[g] = a.gi_row ? gir[g * a.Hp + j] : dd_ldf<COH>(gir + g * a.Hp + j);   // layer 0: the constant g1 table
        }
        if (bi < NB) e_c = dd_ldf<COH>(a.c + (int64_t)b * a.Hp + j);
    } else {
        const int n = 16 * nt + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int bi = m0 + 4 * kq + r;
            rb_out[r] = bi < NB ? real(bi) : 0;
            if constexpr (PRED) nd_out[r] = bi < NB ? dd_ldi<COH>(a.need + rb_out[r]) : 0;
        }
        if (n < a.N) {
            if (a.bias) e_bias = a.bias[n];
            if (EPI == SK_ACT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = rb_out[r];
                    if (m0 + 4 * kq + r < NB) {
                        int tt = dd_ldi<COH>(a.t + b);
                        const int Tb = a.Tb ? a.Tb[b] : a.T;                                   // ragged batch: this utterance's frames / first enc_proj row
                        const int64_t r0 = a.row0 ? (int64_t)a.row0[b] : (int64_t)b * a.T;
                        if constexpr (WF > 1) {
#pragma unroll
                            for (int f = 1; f < WF; ++f) {
                                const int tf = tt + f < Tb ? tt + f : Tb - 1;
                                e_epw[r][f - 1] = f < a.F ? a.ep[(r0 + tf) * a.N + n] : 0.0f;
                            }
                        }
                        tt = tt < Tb ? tt : Tb - 1;
                        e_ep[r] = a.ep[(r0 + tt) * a.N + n];
                    }
                }
            }
        }
    }
    // software pipeline: chunks of 4 float4 pairs (16 MFMAs, ~640 cycles) with the next chunk's loads in flight
    constexpr int CH = 4;
    const int nchunks = a.K / (16 * CH);
    float4 xa[CH], wa[CH], xb[CH], wb[CH];
#define SK_LOAD(X_, W_, c_)                                                         \
    _Pragma("unroll") for (int i = 0; i < CH; ++i) {                               \
        X_[i] = dd_ld4<COH>(xq + 4 * ((c_) * CH + i));                              \
        W_[i] = dd_ldw<NTW>(wq + 4 * ((c_) * CH + i));                               \
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
    // one single-chain product acc = X W^T over K (natural k order) through the two-set prefetch ring
    auto chain = [&](const float4 *xq, const float4 *wq) -> f32x4 {
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (PRED) {
            if (pred_skip) return acc;
        }
        if constexpr (NCH > 0) {
            // two register sets, fully unrolled: chunk c+1 is in flight while chunk c feeds the MFMA chain (>= 640 cycles of cover).
            // A third set hid more latency but pushed the kernel past 96 VGPRs, and then a decode wave no longer fits next to the
            // four 104-VGPR waves per SIMD of the 128x128 GEMM of the NEXT batch's encoder (two-stream pipeline): every decode
            // workgroup had to wait for a GEMM workgroup to retire and then held that slot -- 2.0 ms per 64-clip batch.
            SK_LOAD(xa, wa, 0)
            if constexpr (PRED) {                                    // (the flags were requested first: this waits for them, not for the chunk)
                const bool mine = EPI == SK_CELL ? nd_cell != 0 : (nd_out[0] | nd_out[1] | nd_out[2] | nd_out[3]) != 0;
                if (__builtin_amdgcn_ballot_w64(mine) == 0) { pred_skip = true; return acc; }
            }
#pragma unroll
            for (int c = 0; c < NCH; c += 2) {
                if (c + 1 < NCH) { SK_LOAD(xb, wb, c + 1) }
                SK_MMA(xa, wa)
                if (c + 2 < NCH) { SK_LOAD(xa, wa, c + 2) }
                if (c + 1 < NCH) { SK_MMA(xb, wb) }
            }
        } else {
            SK_LOAD(xa, wa, 0)
            if constexpr (PRED) {
                const bool mine = EPI == SK_CELL ? nd_cell != 0 : (nd_out[0] | nd_out[1] | nd_out[2] | nd_out[3]) != 0;
                if (__builtin_amdgcn_ballot_w64(mine) == 0) { pred_skip = true; return acc; }
            }
            for (int c = 0; c < nchunks; c += 2) {
                if (c + 1 < nchunks) { SK_LOAD(xb, wb, c + 1) }
                SK_MMA(xa, wa)
                if (c + 2 < nchunks) { SK_LOAD(xa, wa, c + 2) }
                if (c + 1 < nchunks) { SK_MMA(xb, wb) }
            }
        }
        return acc;
    };
    // SK_CELL of an upper LSTM layer (a.W2 set): the layer's input projection W_ih x + b_ih (x = the h' of the layer below) is the
    // same kind of chain over the same 16 gate columns -- computed here, before the W_hh chain, instead of in a launch of its own
    // (one launch and one round trip of gi through memory less per step; the two chains and their sum are the per-phase kernels' own:
    // gi = chain_ih + b_ih, gates = gi + chain_hh, src/lstm.cpp:15)
    f32x4 acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (EPI == SK_CELL && a.W2) {
        acc2 = chain(reinterpret_cast<const float4 *>(a.X2 + (int64_t)xrow * a.K) + kq, reinterpret_cast<const float4 *>(a.W2 + (int64_t)wrow * a.K) + kq);
    }
    const f32x4 acc = chain(xp, wp);
    if constexpr (PRED) { if (pred_skip) return; }
#undef SK_LOAD
#undef SK_MMA
    // C/D layout of 16x16x4: column = lane & 15, row (utterance) = 4 * (lane >> 4) + r
    if (EPI == SK_BIAS) {
        const int n = 16 * nt + col;
        if (n < a.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (m0 + 4 * kq + r < NB) dd_stf<COH>(a.out + (int64_t)rb_out[r] * a.ldo + n, a.bias ? acc[r] + e_bias : acc[r]);
            }
        }
    } else if (EPI == SK_ACT) {
        // z = relu(enc_proj(enc_t) + pred_proj(pred) [+ bp])   src/tdt.cpp:17-18 ; written in sigma layout
        const int n = 16 * nt + col;
        if (n < a.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (m0 + 4 * kq + r >= NB) continue;
                if constexpr (PRED) { if (!nd_out[r]) continue; }
                const int b = rb_out[r];
                float p = acc[r];
                if (a.bias) p = p + e_bias;
                if (a.pp_out) a.pp_out[(int64_t)b * a.N + n] = p;           // cached for the steps after a blank (TdtState::pp)
                const float s = e_ep[r] + p;
                if constexpr (WF > 1) {
                    float *zr = a.out + (int64_t)b * a.F * a.N + sigma16(n);
                    dd_stf<COH>(zr, s > 0.0f ? s : 0.0f);
#pragma unroll
                    for (int f = 1; f < WF; ++f) {
                        if (f < a.F) {
                            const float sf = e_epw[r][f - 1] + p;
                            dd_stf<COH>(zr + (int64_t)f * a.N, sf > 0.0f ? sf : 0.0f);
                        }
                    }
                } else {
                    dd_stf<COH>(a.out + (int64_t)b * a.N + sigma16(n), s > 0.0f ? s : 0.0f);
                }
            }
        }
    } else {
        // LSTMCell::forward: gates = (W_ih x + b) + W_hh h ; i,f,g,o ; c' = f*c + i*g ; h' = o*tanh(c')
        const int ul = lane >> 2, jj = lane & 3;
        const int j = 4 * nt + jj;
        const bool row_ok = m0 + ul < NB && (!PRED || nd_cell != 0);
        const int b = rb_cell;
        if (a.W2) {                                   // upper layer: gi = chain_ih + b_ih, through the same LDS transposition
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[wave][4 * kq + r][col] = acc2[r];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (row_ok) {
#pragma unroll
                for (int g = 0; g < 4; ++g) e_gi[g] = tile[wave][ul][4 * g + jj] + a.bias2[g * a.Hp + j];
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[wave][4 * kq + r][col] = acc[r];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (row_ok) {
            const float gi_ = e_gi[0] + tile[wave][ul][jj];
            const float gf_ = e_gi[1] + tile[wave][ul][4 + jj];
            const float gg_ = e_gi[2] + tile[wave][ul][8 + jj];
            const float go_ = e_gi[3] + tile[wave][ul][12 + jj];
            const float ig = dsigmoidf(gi_), fg = dsigmoidf(gf_), gg = dtanhf(gg_), og = dsigmoidf(go_);
            const float t1 = fg * e_c;
            const float t2 = ig * gg;
            const float cnew = t1 + t2;
            dd_stf<COH>(a.cn + (int64_t)b * a.Hp + j, cnew);
            dd_stf<COH>(a.out + (int64_t)b * a.Hp + sigma16(j), og * dtanhf(cnew));   // h' in sigma layout (it is only ever a GEMV operand)
        }
    }
}


// ---- greedy decision of one utterance (src/tdt.cpp:76-105, src/rnnt.cpp:82-107, src/phrase_boost.cpp:177-350) --------------------
struct BestLP {
    float lp;
    int idx;
};
__device__ __forceinline__ BestLP wave_logsoftmax_argmax(const float *__restrict__ x, int n, float *__restrict__ lp_out, int lane) {
    if (n <= 8) {
        // a head of a few values (the TDT durations): lanes 8 .. 63 would carry the identities through the first three steps of every tree -- the last three
        // steps give lanes 0 .. 7 the same bits (pk_devmath.h: wave_sum_low8); lanes >= 8 return what lane (l & 7) returns
        const int i = lane & 7;
        const bool in = i < n;
        const float xi = in ? x[i] : -__builtin_huge_valf();
        const float m = wave_max_low8(xi);
        const float lse = dlogf(wave_sum_low8(in ? dexpf_nonpos(xi - m) : 0.0f));
        float best = in ? (xi - m) - lse : -__builtin_huge_valf();
        int bi = in ? i : 0x7fffffff;
        if (lp_out && in && lane < 8) lp_out[i] = best;
        auto step = [&](auto off) {
            const float ob = wave_xor<decltype(off)::value>(best);
            const int oi = wave_xor_i<decltype(off)::value>(bi);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        };
        step(std::integral_constant<int, 4>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 1>{});
        return {best, bi};
    }
    float m = -__builtin_huge_valf();
    for (int i = lane; i < n; i += 64) m = fmaxf(m, x[i]);
    m = wave_max64(m);
    float p = 0.0f;
    for (int i = lane; i < n; i += 64) p = p + dexpf_nonpos(x[i] - m);
    const float lse = dlogf(wave_sum64(p));
    float best = -__builtin_huge_valf();
    int bi = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const float l = (x[i] - m) - lse;
        if (lp_out) lp_out[i] = l;
        if (bi == 0x7fffffff || l > best) { best = l; bi = i; }
    }
    wave_butterfly([&](auto off) {
        const float ob = wave_xor<decltype(off)::value>(best);
        const int oi = wave_xor_i<decltype(off)::value>(bi);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    });
    return {best, bi};
}


// One utterance's decision of a lock-step decode step, by one 256-thread workgroup; sm = x[V+D], e[V+D], scratch[16] (+ BOOST: mask,
// active sets).  COH: the words other workgroups of the same launch wrote / will read go through system-scope accesses.
// SCORE: the teacher-forced form (pk_tdt_score, TdtState::force_label) -- compiled as its own kernel so that the decode loop's decision
// kernel carries none of it (round 4: with the scoring code inline the V = 8193 decision went from 15.2 to 17.3 us per launch).
// FAST (round 6; round-5 verdict item 6): the tolerance-class mode's plain greedy step (TdtState::h_bf16, no boosting, no forced scoring).  The logits row stays in
// REGISTERS after its one round trip: row maximum, sum of hardware exp2, arg-maximum (of the raw logits: log-softmax is monotone) and runner-up are per-thread
// scans over the registers + ONE cross-wave exchange -- two barriers in all, no LDS sweep over the vocabulary (the exact form: four barriers and three sweeps, the
// canonical single-wave sum64 among them; 17 us per step at vocabulary 8193).  Not bit-identical to the exact form (summation order, exp, ties of ROUNDED
// log-probs): only where the mode is compared within a tolerance.  The duration head and everything behind the decision are the same code.
// NC: 256-element slots of the candidate LSTM state a thread carries in registers (L * Hp <= NC * 256; 3 covers one layer of 640, the launcher picks) -- every slot is
// a guarded load, a guarded store and their address arithmetic, emitted whether the model needs it or not.
template <bool BOOST, bool COH, bool SCORE = false, bool FAST = false, int NC = 12>
__device__ __forceinline__ void tdt_decide_one(const TdtState &st, int b, float *sm) {
    static_assert(!FAST || (!BOOST && !COH && !SCORE), "the fast decision is the plain greedy step of the launch-per-phase loop");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // A finished utterance leaves without a write.  Launch-per-phase loop: the flag travels WITH the step's other loads (state words, candidate LSTM
    // state, logits row) and is looked at once they are all in flight -- as the first statement it was a round trip of its own (~1 us of a ~7 us launch).
    DEC_STAMPS_DECL;
    DEC_STAMP(0);                                                   // entry
    const int done_in = dd_ldi<COH>(st.done + b);
    if constexpr (COH) { if (done_in) return; }
    int n_force = 0;                                               // SCORE: steps of this utterance's given path, first element of its arrays
    int64_t force_off = 0;
    if constexpr (SCORE) {
        n_force = st.n_force_b ? st.n_force_b[b] : st.n_force;
        force_off = (int64_t)b * st.force_stride;
        if (st.force_label && n_force <= 0) {                      // nothing to walk in this chunk: finished before the first decision
            if (done_in) return;
            if (tid == 0) { st.lens[b] = 0; dd_sti<COH>(st.done + b, 1); atomicAdd(st.done_count, 1); if (st.need) dd_sti<COH>(st.need + b, 0); }
            return;
        }
    }
    // ragged batch (TdtState::Tb / row0): this utterance's frame count, first enc_proj row and its own cap on joint evaluations
    const int Tb = st.Tb ? st.Tb[b] : st.T;
    const int64_t ep_row0 = st.row0 ? (int64_t)st.row0[b] : (int64_t)b * st.T;
    const int max_steps_b = (st.Tb && st.max_steps > 0) ? Tb * (st.max_symbols + 1) + 16 : st.max_steps;
    const int VD = st.V + st.D;
    const int Fw = (!BOOST && !COH && !SCORE && st.F > 1) ? st.F : 1;      // frame window (below); its F logits rows sit in front of the scratch
    float *x = sm, *e = sm + (int64_t)Fw * VD;                     // x: the row under decision (window: row f of the F staged rows)
    float *red = e + VD;                                           // [0..3] wave maxima, [4] lse, [8..11] best val, [12..15] best idx
    const int MW = (st.V + 31) >> 5;
    unsigned *mask = reinterpret_cast<unsigned *>(red + 16);       // [MW] boosted-token bits
    int *acts = reinterpret_cast<int *>(mask + MW);                // [kTrieMaxActive] this step's active states
    int *nx = acts + kTrieMaxActive;                               // [1 + kTrieMaxActive] next active set (count first)
    int n_act = 0;
    if constexpr (BOOST) {
        n_act = st.trie.n_act[b];
        if (tid < n_act) acts[tid] = st.trie.act[(int64_t)b * kTrieMaxActive + tid];
        for (int i = tid; i < MW; i += 256) mask[i] = 0u;
    }
    // FRAME WINDOW (TdtState::F > 1, small lock-step batches; plain greedy step only): the heads product evaluated the joint for the F frames t .. t + F - 1 under
    // the unchanged prediction-net state, so a blank decision whose successor frame lies inside the window is followed by the next decision at once -- the
    // evaluations, their order, the counters and the running margin are those of the one-decision-per-launch loop; a run of blanks costs one launch.
    // The F rows are staged in LDS by the first decision's load (one round trip for all of them; launch_tdt_decide sizes the scratch): a walked decision starts from
    // LDS, not from memory.  F > 1 comes with V + D <= 5 x 256 (Model::run_tdt_loop).
    const float *lg = st.logits + (int64_t)b * Fw * VD;
    // issue every independent global load up front (state words, candidate LSTM state): each dependent round trip to
    // L2/HBM costs ~1-2 us in this latency-bound kernel
    const int t_in = dd_ldi<COH>(st.t + b), steps_in = dd_ldi<COH>(st.steps + b), n_out_in = dd_ldi<COH>(st.n_out + b), nsym_in = dd_ldi<COH>(st.nsym + b);
    constexpr int kMaxCarry = NC;                                  // L * Hp <= NC * 256
    float hcar[kMaxCarry], ccar[kMaxCarry];
    const int n_state = st.L * st.Hp;
    // tolerance-class mode (decode_gemv_bf16.hip): h / h' are bf16 arrays [L][B][Hp]; they are committed as Hp / 2 float-sized words per row
    const int hp_h = st.h_bf16 ? st.Hp / 2 : st.Hp, n_h = st.L * hp_h;
    // element i of the [L][width] state of this utterance sits at ((l * B + b) * width + r), i = l * width + r: found by L - 1 compare-and-subtract steps (none for a
    // one-layer net) -- as i / width and i % width these were 48 integer divisions, ~1 500 instructions in front of the loads of a kernel that is all latency
    auto state_off = [&](int i, int width) -> int64_t {
        int l = 0, r = i;
        for (int k = 1; k < st.L; ++k) if (r >= width) { r -= width; ++l; }
        return ((int64_t)l * st.B + b) * width + r;
    };
    if constexpr (!COH) {                                          // (persistent kernel: copied at commit time instead -- 24 fewer live
#pragma unroll                                                     //  registers, it has to fit beside the encoder's GEMM waves)
        for (int q = 0; q < kMaxCarry; ++q) {
            const int i = tid + 256 * q;
            if (i < n_state) ccar[q] = st.cn[state_off(i, st.Hp)];
            if (i < n_h) hcar[q] = st.hn[state_off(i, hp_h)];
        }
    }
    // the running margin's old value and the cached pred_proj row (blank steps form the next z from it): requested here, consumed after the decision
    float mg_old = 0.0f;
    if constexpr (!COH && !BOOST) { if (st.margin && tid == 0) mg_old = st.margin[b]; }
    constexpr int kPpCarry = 4;                                    // J <= 4 * 256
    float ppv[kPpCarry];
    const bool pp_carried = !COH && st.need && st.J <= 256 * kPpCarry;
    if (pp_carried) {
#pragma unroll
        for (int q = 0; q < kPpCarry; ++q) {
            const int n = tid + 256 * q;
            ppv[q] = n < st.J ? st.pp[(int64_t)b * st.J + n] : 0.0f;
        }
    }
    BestLP lab{0.0f, 0};
    float sec = -__builtin_huge_valf();
    int skip = 1;
    int t_cur = t_in, steps_cur = steps_in, nsym_cur = nsym_in, f_cur = 0;     // the walk through the frame window
    float vw[kDecWindowMax][5];                                                 // its F logits rows on their way from memory (exact form, V + D <= 5 x 256)
    for (;;) {
    DEC_STAMP(1);                                                   // a decision starts ([11]: the first one's start, [15]: decisions so far)
#ifdef DEC_TRACE
    if (f_cur == 0) dec_st[11] = dec_st[1];
    dec_st[15] += 1;
#endif
    if constexpr (FAST) {
      auto fast = [&](auto nq_tag) {
        constexpr int NQ = decltype(nq_tag)::value;                 // VD <= NQ x 256 (33: launch_tdt_decide checks)
        float v[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int i = tid + 256 * u;
            v[u] = lg[i < VD ? i : VD - 1];
        }
        if (done_in) return true;
        float mx = -__builtin_huge_valf(), s2 = -__builtin_huge_valf();
        int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {                              // maximum, first arg-maximum and runner-up of this thread's values (increasing index)
            const int i = tid + 256 * u;
            if (i < st.V) {
                if (v[u] > mx) { s2 = mx; mx = v[u]; bi = i; }
                else s2 = fmaxf(s2, v[u]);
            } else if (i < VD) {
                x[i] = v[u];                                        // the few duration logits: wave 1 reads them from LDS below
            }
        }
        wave_butterfly([&](auto off) {
            constexpr int O = decltype(off)::value;
            const float ob = wave_xor<O>(mx), os = wave_xor<O>(s2);
            const int oi = wave_xor_i<O>(bi);
            const bool take = ob > mx || (ob == mx && oi < bi);
            s2 = fmaxf(fmaxf(s2, os), take ? mx : ob);
            if (take) { mx = ob; bi = oi; }
        });
        if (lane == 0) { red[wave] = mx; red[8 + wave] = s2; red[12 + wave] = __int_as_float(bi); }
        __syncthreads();
        float m = red[0], s2b = red[8];
        int bib = __float_as_int(red[12]);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ob = red[w], os = red[8 + w];
            const int oi = __float_as_int(red[12 + w]);
            const bool take = ob > m || (ob == m && oi < bib);
            s2b = fmaxf(fmaxf(s2b, os), take ? m : ob);
            if (take) { m = ob; bib = oi; }
        }
        float p = 0.0f;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int i = tid + 256 * u;
            if (i < st.V) p += __builtin_amdgcn_exp2f((v[u] - m) * 1.44269502162933349609375f);
        }
        p = wave_sum64(p);
        if (lane == 0) red[4 + wave] = p;                           // (red[4 .. 7]: the waves' partial sums)
        if (wave == 1 && st.D > 0) {                                // duration head: a few values, one wavefront (the exact form's code)
            const BestLP dur = wave_logsoftmax_argmax(x + st.V, st.D, e + st.V, lane);
            if (lane == 0) e[0] = (float)(dur.idx < st.D ? st.durations[dur.idx] : 1);          // (e[0 .. V) is free in this form: the skip, the duration margin)
            if (st.margin) {
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                float second = -__builtin_huge_valf();
                for (int i = lane; i < st.D; i += 64)
                    if (i != dur.idx) second = fmaxf(second, e[st.V + i]);
                second = st.D <= 8 ? wave_max_low8(second) : wave_max64(second);   // (lane 0 reads it; D <= 8: the same bits from the last three steps)
                if (lane == 0) e[1] = st.D > 1 ? dur.lp - second : __builtin_huge_valf();
            }
        }
        __syncthreads();
        const float lse = dlogf((red[4] + red[5]) + (red[6] + red[7]));
        lab.idx = bib;
        lab.lp = 0.0f - lse;                                        // (x_best - m) - lse with x_best == m
        sec = (s2b - m) - lse;
        if (st.margin && tid == 0) {
            float mg = lab.lp - sec;
            if (st.D > 0) mg = fminf(mg, e[1]);
            mg_old = mg < mg_old ? mg : mg_old;
        }
        if (st.D > 0) skip = (int)e[0];
        return false;
      };
      if (VD <= 256 * 5) { if (fast(std::integral_constant<int, 5>{})) return; }
      else if (fast(std::integral_constant<int, 33>{})) return;
    } else {
    float m = -__builtin_huge_valf();
    // The logits row with ALL of a thread's loads in flight at once (one L2 / HBM round trip): written as a plain loop, every iteration
    // waited for its own trip -- 33 dependent trips at vocabulary 8193, 26 us for ~3 us of arithmetic (round 3).  Rows longer than 33 x 256
    // (and the register-capped persistent kernel) go in batches of 8.
    auto stage_row = [&](auto nq_tag) {
        constexpr int NQ = decltype(nq_tag)::value;
        float v[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int i = tid + 256 * u;
            v[u] = dd_ldf<COH>(lg + (i < VD ? i : VD - 1));
        }
        if (done_in) return true;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int i = tid + 256 * u;
            if (i < VD) {
                x[i] = v[u];
                if (i < st.V) m = fmaxf(m, v[u]);
            }
        }
        return false;
    };
    if (Fw > 1) {
        if (f_cur == 0) {
            // row 0 first, the other rows behind it in the same queue: the first decision waits for ITS row only (loads return in order), the rest arrive under it
            // and go to LDS when -- if -- the walk takes its first step
#pragma unroll
            for (int f = 0; f < kDecWindowMax; ++f)
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int i = tid + 256 * u;
                    vw[f][u] = f < Fw ? lg[(int64_t)f * VD + (i < VD ? i : VD - 1)] : 0.0f;
                }
            if (done_in) return;
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int i = tid + 256 * u;
                if (i < VD) sm[i] = vw[0][u];
                if (i < st.V) m = fmaxf(m, vw[0][u]);
            }
        } else {
            x = sm + (int64_t)f_cur * VD;                           // staged by the first decision, barriers since
#pragma unroll
            for (int u = 0; u < 5; ++u) { const int i = tid + 256 * u; if (i < st.V) m = fmaxf(m, x[i]); }
        }
    } else if (!COH && VD <= 256 * 5) {
        if (stage_row(std::integral_constant<int, 5>{})) return;
    } else if (!COH && VD <= 256 * 33) {
        if (stage_row(std::integral_constant<int, 33>{})) return;
    } else {
        if (done_in) return;
        for (int i0 = tid; i0 < VD; i0 += 256 * 8) {
            float v8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u;
                v8[u] = dd_ldf<COH>(lg + (i < VD ? i : VD - 1));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 256 * u;
                if (i < VD) {
                    x[i] = v8[u];
                    if (i < st.V) m = fmaxf(m, v8[u]);
                }
            }
        }
    }
    DEC_STAMP(2);                                                   // row arrived, staged
    m = wave_max64(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    DEC_STAMP(3);                                                   // maximum known
    for (int i0 = tid; i0 < st.V; i0 += 256 * 4) {                 // (4 independent LDS reads / exp chains per trip)
        float t4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + 256 * u; t4[u] = x[i < st.V ? i : st.V - 1]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + 256 * u; if (i < st.V) e[i] = dexpf_nonpos(t4[u] - m); }
    }
    if constexpr (BOOST) {                                         // get_boosted_tokens: union of the children of the active states
        for (int a = 0; a < n_act; ++a) {
            const int sn = acts[a];
            const int c1 = st.trie.off[sn + 1];
            for (int c = st.trie.off[sn] + tid; c < c1; c += 256) {
                const int tk = st.trie.tok[c];
                if (tk >= 0 && tk < st.V) atomicOr(&mask[tk >> 5], 1u << (tk & 31));
            }
        }
    }
    __syncthreads();
    DEC_STAMP(4);                                                   // exps in LDS
    if (wave == 0) {
        float p = 0.0f;
        for (int i0 = lane; i0 < st.V; i0 += 64 * 8) {              // the canonical strided partial sum, its LDS reads 8 at a time; the adds stay in index order
            float t8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u; t8[u] = e[i < st.V ? i : st.V - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u; if (i < st.V) p = p + t8[u]; }
        }
        const float lse = dlogf(wave_sum64(p));
        if (lane == 0) red[4] = lse;
    }
    if (wave == 1 && st.D > 0) {                                   // duration head: a few values, one wavefront
        const BestLP dur = wave_logsoftmax_argmax(x + st.V, st.D, e + st.V, lane);      // (log-probs into the free tail of e[]: read back for the margin)
        if (lane == 0) red[5] = (float)(dur.idx < st.D ? st.durations[dur.idx] : 1);
        if (st.margin) {                                           // top-1 / top-2 margin of the duration decision (it moves the frame pointer)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            float second = -__builtin_huge_valf();
            for (int i = lane; i < st.D; i += 64)
                if (i != dur.idx) second = fmaxf(second, e[st.V + i]);
            second = st.D <= 8 ? wave_max_low8(second) : wave_max64(second);   // (lane 0 reads it; D <= 8: the same bits from the last three steps)
            if (lane == 0) red[6] = st.D > 1 ? dur.lp - second : __builtin_huge_valf();
        }
    }
    DEC_STAMP(5);                                                   // wave 0: sum, log done (before the barrier)
    __syncthreads();
    DEC_STAMP(6);                                                   // lse known (duration head done)
    const float lse = red[4];
    float best = -__builtin_huge_valf();
    // `second` = the largest label log-prob that is NOT the winner's (ties with the winner count: margin 0).  It only feeds the optional
    // top-1 / top-2 margin report (st.margin); the decision itself is the first-maximum argmax exactly as before.
    float second = -__builtin_huge_valf();
    int bi = 0x7fffffff;
    for (int i0 = tid; i0 < st.V; i0 += 256 * 4) {
        float t4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + 256 * u; t4[u] = x[i < st.V ? i : st.V - 1]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            if (i < st.V) {
                float l = (t4[u] - m) - lse;
                if constexpr (BOOST) l = l + (((mask[i >> 5] >> (i & 31)) & 1u) ? st.trie.boost : 0.0f);
                if (bi == 0x7fffffff || l > best) { second = best; best = l; bi = i; }
                else second = fmaxf(second, l);
            }
        }
    }
    wave_butterfly([&](auto off) {
        constexpr int O = decltype(off)::value;
        const float ob = wave_xor<O>(best);
        const int oi = wave_xor_i<O>(bi);
        const float os = wave_xor<O>(second);
        const bool take = ob > best || (ob == best && oi < bi);
        second = fmaxf(fmaxf(second, os), (oi == bi) ? -__builtin_huge_valf() : (take ? best : ob));   // the loser's maximum joins the rest
        if (take) { best = ob; bi = oi; }
    });
    if (lane == 0) { red[8 + wave] = best; red[12 + wave] = __int_as_float(bi); red[wave] = second; }   // (red[0..3]: the wave maxima were consumed two barriers ago)
    __syncthreads();
    DEC_STAMP(7);                                                   // argmax exchanged
    lab = BestLP{red[8], __float_as_int(red[12])};
    sec = red[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float ob = red[8 + w];
        const int oi = __float_as_int(red[12 + w]);
        const bool take = ob > lab.lp || (ob == lab.lp && oi < lab.idx);
        sec = fmaxf(fmaxf(sec, red[w]), (oi == lab.idx) ? -__builtin_huge_valf() : (take ? lab.lp : ob));
        if (take) { lab.lp = ob; lab.idx = oi; }
    }
    if constexpr (!BOOST) {
        if (st.margin && tid == 0) {                               // running minimum over the utterance's decisions (SURVEY 8c early warning)
            float mg = lab.lp - sec;
            if (st.D > 0) mg = fminf(mg, red[6]);
            if constexpr (COH) {
                const float old = dd_ldf<COH>(st.margin + b);
                dd_stf<COH>(st.margin + b, mg < old ? mg : old);
            } else {
                mg_old = mg < mg_old ? mg : mg_old;                  // (written once, behind the walk)
            }
        }
    }
    if constexpr (BOOST) lab.lp = (x[lab.idx] - m) - lse;          // the confidence is the UNBOOSTED log-prob (phrase_boost.cpp:313-315)
    if (st.D > 0) skip = (int)red[5];
    if constexpr (SCORE && !BOOST) {
        if (st.score_lab) {                                        // teacher-forced scoring: the label log-prob row as the joint returns it (its own pass:
            float *row = st.score_lab + (force_off + steps_cur) * st.V;   // nothing of it sits in the decode loop's argmax sweep)
            for (int i = tid; i < st.V; i += 256) row[i] = (x[i] - m) - lse;
        }
        if (st.force_label) {                                      // teacher-forced scoring (TdtState::force_label): the given decision, not the argmax
            if (st.score_dur && tid < st.D) st.score_dur[(force_off + steps_cur) * st.D + tid] = e[st.V + tid];
            const int k = steps_cur < n_force ? steps_cur : n_force - 1;
            lab.idx = st.force_label[force_off + k];
            lab.lp = (x[lab.idx] - m) - lse;
            if (st.D > 0) skip = st.durations[st.force_dur[force_off + k]];
        }
    }
    }                                                               // (exact form)
    if (Fw > 1 && lab.idx == st.blank) {                            // blank inside the window: the next frame's logits row is already there
        const int adv = (st.D > 0) ? (skip > 1 ? skip : 1) : 1;
        const bool capped = max_steps_b > 0 && steps_cur + 1 >= max_steps_b;
        if (f_cur + adv < Fw && t_cur + adv < Tb && !capped) {
            if constexpr (!FAST) {
                if (f_cur == 0) {                                   // first step of the walk: the rows behind row 0 leave the registers
#pragma unroll
                    for (int f = 1; f < kDecWindowMax; ++f)
#pragma unroll
                        for (int u = 0; u < 5; ++u) {
                            const int i = tid + 256 * u;
                            if (f < Fw && i < VD) sm[(int64_t)f * VD + i] = vw[f][u];
                        }
                }
            }
            f_cur += adv; t_cur += adv; ++steps_cur; nsym_cur = 0;
            __syncthreads();                                        // the decision scratch (x, e, red) is rewritten
            continue;
        }
    }
    break;
    }                                                               // (window walk)
    if constexpr (!COH && !BOOST) { if (st.margin && tid == 0) st.margin[b] = mg_old; }
    DEC_STAMP(8);                                                   // walk over
    const int lane0 = tid;                                         // thread 0 writes the scalar state
    // scalar control (wave-uniform values; lane 0 writes)
    int t = t_cur;
    const int nsteps = steps_cur + 1;
    int n_out = n_out_in;
    int nsym = nsym_cur;
    const bool commit = lab.idx != st.blank;
    if (!commit) {
        // blank: the LSTM state reverts -- the candidates hn/cn are simply not committed (src/tdt.cpp:88-93)
        t += (st.D > 0) ? (skip > 1 ? skip : 1) : 1;
        nsym = 0;
    } else {
        if (lane0 == 0) {
            if (n_out < st.max_tokens) {
                const int64_t o = (int64_t)b * st.max_tokens + n_out;
                st.ids[o] = lab.idx;
                st.start[o] = t;
                int e = st.D > 0 ? t + (skip > 1 ? skip : 1) - 1 : t;     // src/tdt.cpp:184-187 ; rnnt.cpp:170 (end = t)
                st.end[o] = (st.keep_state || e < Tb) ? e : Tb - 1;
                st.conf[o] = dexpf(lab.lp);                                 // confidence = exp(max log-prob) :169
            }
            dd_sti<COH>(st.token + b, lab.idx);
        }
        if constexpr (BOOST) {                        // ContextTrie::advance on the emitted token (phrase_boost.cpp:52-66, :336)
            if (tid == 0) { nx[0] = 1; nx[1] = 0; }   // the root is always active
            __syncthreads();
            for (int a = 0; a < n_act; ++a) {
                const int sn = acts[a];
                const int c1 = st.trie.off[sn + 1];
                for (int c = st.trie.off[sn] + tid; c < c1; c += 256)
                    if (st.trie.tok[c] == lab.idx) {
                        const int slot = atomicAdd(&nx[0], 1);
                        if (slot < kTrieMaxActive) nx[1 + slot] = st.trie.node[c];
                    }
            }
            __syncthreads();
            const int nn = nx[0] < kTrieMaxActive ? nx[0] : kTrieMaxActive;
            if (tid < nn) st.trie.act[(int64_t)b * kTrieMaxActive + tid] = nx[1 + tid];
            if (tid == 0) st.trie.n_act[b] = nn;
        }
        ++n_out;
        if (st.D > 0) {
            if (skip > 0) t += skip;                  // duration 0: emit another symbol on the same frame (:99-105)
        } else if (++nsym >= st.max_symbols) {        // RNNT: the inner for runs out -> next frame (rnnt.cpp:82-107)
            t += 1;
            nsym = 0;
        }
        if constexpr (COH) {                          // commit the candidate LSTM state, [L][B][Hp]
            for (int l = 0; l < st.L; ++l) {
                const int64_t base = ((int64_t)l * st.B + b) * st.Hp;
#pragma unroll 2
                for (int j = tid; j < st.Hp; j += 256) {
                    dd_stf<true>(st.h + base + j, dd_ldf<true>(st.hn + base + j));
                    dd_stf<true>(st.c + base + j, dd_ldf<true>(st.cn + base + j));
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < kMaxCarry; ++q) {
                const int i = tid + 256 * q;
                if (i < n_state) st.c[state_off(i, st.Hp)] = ccar[q];
                if (i < n_h) st.h[state_off(i, hp_h)] = hcar[q];
            }
        }
    }
    if (st.need) {
        // prediction-net caching (TdtState::need): a token changed (token, h, c) -> the next step runs the cells and pred_proj again; a blank
        // changed only the frame -> this workgroup forms next step's z = relu(enc_proj[t'] + pp) from the cached pp (SK_ACT's epilogue, same
        // operand order: enc_proj + (pred_proj [+ bias]))
        const bool fin = t >= Tb || (max_steps_b > 0 && nsteps >= max_steps_b) || (SCORE && st.force_label && nsteps >= n_force);
        if (tid == 0) dd_sti<COH>(st.need + b, (commit && !fin) ? 1 : 0);
        if (!commit && !fin) {
            const float *epr = st.ep + (ep_row0 + t) * st.J, *ppr = st.pp + (int64_t)b * st.J;
            if (pp_carried) {
                // the rows of the next window: frames t .. t + F - 1 (clamped as SK_ACT clamps).  EVERY load first, then the stores: row by row the loads of row
                // f + 1 waited behind the stores of row f (the pointers may alias as far as the compiler knows) -- eight dependent round trips, ~8 us
                float epv[kDecWindowMax][kPpCarry];
#pragma unroll
                for (int fw = 0; fw < kDecWindowMax; ++fw) {
                    if (fw < Fw) {
                        const int tt = t + fw < Tb ? t + fw : Tb - 1;
                        const float *er = st.ep + (ep_row0 + tt) * st.J;
#pragma unroll
                        for (int q = 0; q < kPpCarry; ++q) { const int n = tid + 256 * q; epv[fw][q] = n < st.J ? er[n] : 0.0f; }
                    }
                }
#pragma unroll
                for (int fw = 0; fw < kDecWindowMax; ++fw) {
                    if (fw < Fw) {
                        const int64_t zr = ((int64_t)b * Fw + fw) * st.J;
#pragma unroll
                        for (int q = 0; q < kPpCarry; ++q) {
                            const int n = tid + 256 * q;
                            if (n < st.J) {
                                const float sv = epv[fw][q] + ppv[q];
                                const float zv = sv > 0.0f ? sv : 0.0f;
                                if (st.h_bf16) reinterpret_cast<__bf16 *>(st.z)[zr + n] = (__bf16)zv;
                                else dd_stf<COH>(st.z + zr + sigma16(n), zv);
                            }
                        }
                    }
                }
            } else
            for (int n = tid; n < st.J; n += 256) {
                const float sv = epr[n] + ppr[n];
                const float zv = sv > 0.0f ? sv : 0.0f;
                if (st.h_bf16) reinterpret_cast<__bf16 *>(st.z)[(int64_t)b * st.J + n] = (__bf16)zv;
                else dd_stf<COH>(st.z + (int64_t)b * st.J + sigma16(n), zv);
            }
        }
    }
    DEC_STAMP(9);                                                   // commit / next z issued
    if (lane0 == 0) {
        bool finished = t >= Tb || (SCORE && st.force_label && nsteps >= n_force);
        int len = n_out < st.max_tokens ? n_out : st.max_tokens;
        if (!finished && max_steps_b > 0 && nsteps >= max_steps_b) { finished = true; len = -1; }   // safety cap
        dd_sti<COH>(st.t + b, t);
        dd_sti<COH>(st.steps + b, nsteps);
        dd_sti<COH>(st.n_out + b, n_out);
        dd_sti<COH>(st.nsym + b, nsym);
        if (finished) {
            st.lens[b] = len;
            dd_sti<COH>(st.done + b, 1);
            if constexpr (COH) __hip_atomic_fetch_add(st.done_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            else atomicAdd(st.done_count, 1);
        }
    }
    DEC_STAMP(10);                                                  // state words issued
    DEC_STAMPS_FLUSH();
}

}  // namespace pk
