// parakeet.cpp_amd/csrc/kernels/decode_gemv_bf16.hip -- the per-step products of the TDT / RNNT loop in the TOLERANCE-class mode
// (pk_config.gemm_bf16): bf16 operands, fp32 accumulation on v_mfma_f32_16x16x32_bf16.
//
// Why (round 3, profiles/r03_decode_overlap_ab.txt, r03_m1_kernel_stats_600m_bf16.md): with the 8193-entry vocabulary of tdt-600m the fp32
// decode GEMVs (decode_gemv.hip) stream 42 MB of prediction-net and joint weights through L2 per symbol step and are 29 % of the GPU time of
// the bf16 configuration -- next to an encoder whose products run at bf16 rates.  Here the weights are bf16 (half the bytes), the activations
// that are only ever GEMV operands (h, h', z) are stored as bf16, and a K = 640 product is 20 MFMAs instead of 160.
//
// Specification (= the oracle's gemm_bf16 mode, oracle/pk_oracle.c predict_step / joint_hidden): W_hh, W_ih of the upper LSTM layers,
// pred_proj and the label / duration heads are rounded to bf16 once; the layer-0 input projection stays the fp32 table g1 = W_ih0 E + b;
// h' = bf16(o * tanh(c')) (c stays fp32), z = bf16(relu(enc_proj[t] + pred_proj h' [+ b])); every product accumulates in fp32.
// Accumulation ORDER differs from the oracle's k-ordered chain (MFMA blocks of 32 k): compared within the mode's tolerance.
//
// Tiling as decode_gemv.hip: one wavefront per 16 (utterances) x 16 (outputs) tile, a workgroup = one 16-output tile x up to four
// 16-utterance tiles; lane (row / column l & 15, quarter l >> 4) loads the 8 consecutive k of every 32-k block at 8 (l >> 4) with one 16-byte
// load per operand per MFMA -- activations in their natural layout, weights TILED in that load order at upload (engine.cpp upload_dec16: one
// 1 KB block per (tile, 32-k block), a wave's load = 1 KB of consecutive addresses).  Epilogues: LSTM cell (gates -> c', h'), joint activation, bias.
#include "decode_dev.hpp"

namespace pk {

typedef float db_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 db_bf16x8 __attribute__((ext_vector_type(8)));

// MODE / CHK: chosen by the launcher from the arguments (0 every row, 1 need flags as predicates, 2 compacted row list; CHK: padded grid), as in decode_gemv.hip -- a branch on
// an argument in front of the kernel's argument loads splits them into dependent pieces.
template <int EPI, int MODE = 0, bool CHK = false>
__global__ __launch_bounds__(256) void skinny_gemm_bf16_kernel(SkinnyArgs a) {
    __shared__ float tile[4][16][17];
    const int nt = blockIdx.x, mgroup = blockIdx.y;
    // grid.x is the tile count rounded up to a multiple of 8: workgroup id % 8 = XCD, so XCD x owns the output tiles nt % 8 == x of EVERY
    // utterance group and re-reads only its eighth of W (1.3 of the 10.5 MB of the 8198-row heads) from its own L2 step after step
    if constexpr (CHK) {
        if ((EPI == SK_CELL ? 4 : 16) * nt >= (EPI == SK_CELL ? a.Hp : a.N)) return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int m0 = (mgroup * 4 + wave) * 16;
    // prediction-net caching (TdtState::need): the launch covers only the utterances whose flag is set, compacted in ascending order
    __shared__ int lst[kMaxListRows];
    __shared__ int wtot[4];
    int NB = a.B;
    // One row tile (B <= 16): the flags are predicates -- requested with the launch's other operands, looked at once the first operand chunk is under way, applied
    // to the stores -- instead of a list whose round trip sits in front of every other load (decode_dev.hpp: PRED, the fp32 kernels' form of the same thing).
    constexpr bool pred = MODE == 1, listed = MODE == 2;
    if (listed) NB = dd_build_rowlist<false>(a.need, a.B, lst, wtot);
    auto real = [&](int i) { return listed ? lst[i] : i; };
    int nd_cell = 1, nd_out[4] = {1, 1, 1, 1};
    bool pred_skip = false;
    if constexpr (pred) {
        if (wave) return;                                        // one row tile: wave 0 has it
    } else {
        if (m0 >= NB) return;                                    // whole wave out of range (uniform)
    }
    const __bf16 *X = reinterpret_cast<const __bf16 *>(a.X);
    // W (and W2) are TILED in this kernel's load order (engine.cpp upload_dec16): tile nt = nblk consecutive 1 KB blocks [lane][8 bf16];
    // SK_CELL: the tile's columns are the (gate, unit) pairs g * Hp + 4 nt + j
    const int64_t wtile = (int64_t)nt * (a.K / 32) * 64 + lane;
    int xrow = m0 + col;
    xrow = real(xrow < NB ? xrow : NB - 1);
    // epilogue operands first (token -> g1 row, c, enc_proj[t_b], bias): their round trips hide under the MFMA chain
    float e_gi[4] = {0.0f, 0.0f, 0.0f, 0.0f}, e_c = 0.0f;
    float e_ep[4] = {0.0f, 0.0f, 0.0f, 0.0f}, e_bias = 0.0f;
    int rb_cell = 0, rb_out[4] = {0, 0, 0, 0};                   // utterances of this lane's epilogue rows
    if (EPI == SK_CELL) {
        const int bi = m0 + (lane >> 2), j = 4 * nt + (lane & 3);
        const int b = bi < NB ? real(bi) : 0;
        rb_cell = b;
        if (pred) nd_cell = bi < NB ? a.need[b] : 0;
        if (bi < NB && !a.W2) {
            const float *gir = a.gi + (int64_t)(a.gi_row ? a.gi_row[b] : b) * a.gi_ld;
#pragma unroll
            for (int g = 0; g < 4; ++g) e_gi[g] = gir[g * a.Hp + j];
        }
        if (bi < NB) e_c = a.c[(int64_t)b * a.Hp + j];
    } else {
        const int n = 16 * nt + col;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int bi = m0 + 4 * kq + r;
            rb_out[r] = bi < NB ? real(bi) : 0;
            if (pred) nd_out[r] = bi < NB ? a.need[rb_out[r]] : 0;
        }
        if (n < a.N) {
            if (a.bias) e_bias = a.bias[n];
            if (EPI == SK_ACT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = rb_out[r];
                    if (m0 + 4 * kq + r < NB) {
                        int tt = a.t[b];
                        const int Tb = a.Tb ? a.Tb[b] : a.T;                                   // ragged batch (SkinnyArgs::Tb / row0)
                        const int64_t r0 = a.row0 ? (int64_t)a.row0[b] : (int64_t)b * a.T;
                        tt = tt < Tb ? tt : Tb - 1;
                        e_ep[r] = a.ep[(r0 + tt) * a.N + n];
                    }
                }
            }
        }
    }
    // one product acc = X W^T over K: chunks of 4 MFMAs (128 k), the next chunk's 8 loads in flight under the current chunk
    constexpr int CH = 4;
    auto chain = [&](const __bf16 *xr, const void *wt) -> db_f32x4 {
        db_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if (pred_skip) return acc;
        const db_bf16x8 *xq = reinterpret_cast<const db_bf16x8 *>(xr) + kq;                 // 32-k block i at [4 i]
        const db_bf16x8 *wq = reinterpret_cast<const db_bf16x8 *>(wt) + wtile;              // 32-k block i at [64 i]
        const int nblk = a.K / 32;
        db_bf16x8 xa[CH], wa[CH], xb[CH], wb[CH];
        auto load = [&](db_bf16x8 (&x_)[CH], db_bf16x8 (&w_)[CH], int c0) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int blk = c0 + i < nblk ? c0 + i : nblk - 1;
                x_[i] = xq[4 * blk];
                w_[i] = wq[64 * blk];
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mma = [&](const db_bf16x8 (&x_)[CH], const db_bf16x8 (&w_)[CH], int c0) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if (c0 + i < nblk) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x_[i], w_[i], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        load(xa, wa, 0);
        if (pred) {
            const bool mine = EPI == SK_CELL ? nd_cell != 0 : (nd_out[0] | nd_out[1] | nd_out[2] | nd_out[3]) != 0;
            if (__builtin_amdgcn_ballot_w64(mine) == 0) { pred_skip = true; return acc; }
        }
        for (int c0 = 0; c0 < nblk; c0 += 2 * CH) {
            if (c0 + CH < nblk) load(xb, wb, c0 + CH);
            mma(xa, wa, c0);
            if (c0 + 2 * CH < nblk) load(xa, wa, c0 + 2 * CH);
            if (c0 + CH < nblk) mma(xb, wb, c0 + CH);
        }
        return acc;
    };
    db_f32x4 acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (EPI == SK_CELL && a.W2) {                                // upper LSTM layer: its input projection W_ih h'(l-1) + b_ih, same tile columns
        acc2 = chain(reinterpret_cast<const __bf16 *>(a.X2) + (int64_t)xrow * a.K, a.W2);
    }
    const db_f32x4 acc = chain(X + (int64_t)xrow * a.K, a.W);
    if (pred_skip) return;
    // C/D layout of 16x16: column = lane & 15, row (utterance) = 4 * (lane >> 4) + r
    if (EPI == SK_BIAS) {
        const int n = 16 * nt + col;
        if (n < a.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (m0 + 4 * kq + r < NB) a.out[(int64_t)rb_out[r] * a.ldo + n] = a.bias ? acc[r] + e_bias : acc[r];
        }
    } else if (EPI == SK_ACT) {
        // z = relu(enc_proj(enc_t) + pred_proj(pred) [+ bp])   src/tdt.cpp:17-18 ; stored as bf16 (it is only ever the heads' operand)
        const int n = 16 * nt + col;
        if (n < a.N) {
            __bf16 *z = reinterpret_cast<__bf16 *>(a.out);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (m0 + 4 * kq + r >= NB || !nd_out[r]) continue;
                const int b = rb_out[r];
                float p = acc[r];
                if (a.bias) p = p + e_bias;
                if (a.pp_out) a.pp_out[(int64_t)b * a.N + n] = p;           // cached for the steps after a blank (TdtState::pp)
                const float s = e_ep[r] + p;
                z[(int64_t)b * a.N + n] = (__bf16)(s > 0.0f ? s : 0.0f);
            }
        }
    } else {
        // LSTMCell::forward: gates = (W_ih x + b) + W_hh h ; i,f,g,o ; c' = f*c + i*g ; h' = o*tanh(c')
        const int ul = lane >> 2, jj = lane & 3;
        const int j = 4 * nt + jj;
        const bool row_ok = m0 + ul < NB && nd_cell != 0;
        const int b = rb_cell;
        if (a.W2) {
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
            a.cn[(int64_t)b * a.Hp + j] = cnew;
            reinterpret_cast<__bf16 *>(a.out)[(int64_t)b * a.Hp + j] = (__bf16)(og * dtanhf(cnew));   // h' as bf16: it is only ever a GEMV operand
        }
    }
}

template <int EPI>
static void launch_skinny_bf16_epi(const SkinnyArgs &a, dim3 grid, hipStream_t s) {
    const int n_tiles = EPI == SK_CELL ? a.Hp / 4 : (a.N + 15) / 16;
    const bool chk = n_tiles % 8 != 0;
    const int mode = a.need ? (a.B <= 16 ? 1 : 2) : 0;
#define PK_SB16(M_) do { if (chk) hipLaunchKernelGGL((skinny_gemm_bf16_kernel<EPI, M_, true>), grid, dim3(256), 0, s, a); \
                         else hipLaunchKernelGGL((skinny_gemm_bf16_kernel<EPI, M_, false>), grid, dim3(256), 0, s, a); } while (0)
    if (mode == 1) PK_SB16(1);
    else if (mode == 2) PK_SB16(2);
    else PK_SB16(0);
#undef PK_SB16
}

void launch_skinny_gemm_bf16(const SkinnyArgs &a, int epi, hipStream_t s) {
    const int n_tiles = epi == SK_CELL ? a.Hp / 4 : (a.N + 15) / 16;
    dim3 grid((n_tiles + 7) & ~7, (a.B + 63) / 64);
    switch (epi) {
    case SK_BIAS: launch_skinny_bf16_epi<SK_BIAS>(a, grid, s); break;
    case SK_ACT: launch_skinny_bf16_epi<SK_ACT>(a, grid, s); break;
    case SK_CELL: launch_skinny_bf16_epi<SK_CELL>(a, grid, s); break;
    default: break;
    }
}

}  // namespace pk
