// parakeet.cpp_amd/csrc/kernels/gemm_pipe.hpp -- software-pipelined fp32 MFMA GEMM (gfx950).
//
// out[M][N] = epi(A[M][K] * W[N][K]^T + bias), natural-k fma chains (bit-identical to the oracle's scalar chain).
// Same arithmetic as the first-generation kernel (gemm.hip), rebuilt around the LDS pipe and the per-tile fixed costs:
//  * K inside a BK-wide tile is stored PERMUTED in LDS: position p = (BK/2)*(k&1) + (k>>1).  The 32x32x2 MFMA takes
//    k = 2s from lanes 0-31 and k = 2s+1 from lanes 32-63, so lane (row, h) needs k = h, 2+h, 4+h, ...: with the
//    permutation those are BK/2 CONSECUTIVE floats, fetched as ds_read_b128 (one per 4 MFMA k-steps) instead of one
//    ds_read_b32 per step.  Accumulation order is untouched (step s still consumes k = 2s then 2s+1).
//  * Row pitch BK+4 floats: b128 fragment reads are conflict-free for the hardware's 16-lane groups; the staging
//    stores are ds_write_b64 pairs ({k,k+2} / {k+1,k+3}) with an 8-row interleave that keeps them conflict-free too.
//  * Fragments are double-buffered in registers and the LDS tiles are double-buffered: the reads of sub-step s+1 are
//    issued before the 4*TM*TN MFMAs of sub-step s, the global loads of K tile k+2 are issued a whole K tile before
//    they are stored, and the single barrier per K tile sits before the LAST sub-step, so the first fragment reads
//    of the next K tile are already in flight while this tile's last MFMAs run.
//  * Epilogue through LDS: the accumulators are re-read row-major so that every thread finishes 4 consecutive columns
//    (16-byte stores / residual loads).  Measured on MI355X (profiles/r01_gemm_*): the main loop runs at ~88 % of the
//    MFMA rate; what is left is the write burst of the output tile (all workgroups of a round finish together).
#ifndef PK_GEMM_PIPE_HPP
#define PK_GEMM_PIPE_HPP
#include <cstdio>
#include <cstdlib>
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

typedef float gp_f32x16 __attribute__((ext_vector_type(16)));

// Epilogue shared by the GEMM kernels of this directory (bias, ReLU, SiLU, residual + alpha*y, GLU, sigma column layout) on the
// accumulators of a WGM x WGN grid of waves, each holding TM x TN 32x32 tiles.  `smem` is the kernel's staging memory (free by now),
// CAP its size in floats: the wide path turns the C tile row-major through it (in row bands when it does not fit).
// RS_PER_PASS: the residual rows of a band are requested at the start of that band's pass instead of all up front (tiles of 256 rows:
// NCH = 32 chunks per thread would otherwise hold 128 VGPRs of residual beside the accumulators).
// EXACT (round 5): the caller is an fp32-contract kernel (gemm_pipe_kernel: GemmArgs::fast_act and out_bf16 are never set, the launcher checks) --
// the two run-time switches become compile-time false, which removes a branch and the hardware-exp twin of the activation from every group of four
// results of the fully unrolled epilogue (the bf16 register epilogue lost a third of its clocks to exactly this: gemm_bf16_glds.hpp).
template <int WGM, int WGN, int TM, int TN, int EPI, int CAP_FLOATS, bool RS_PER_PASS = false, bool EXACT = false>
__device__ __forceinline__ void gp_epilogue(const GemmArgs &g_, gp_f32x16 (&acc)[TM][TN], float *smem, int m0, int n0) {
    struct Flags { int fast_act, out_bf16; };
    const Flags gf{EXACT ? 0 : g_.fast_act, EXACT ? 0 : g_.out_bf16};
    const GemmArgs &g = g_;
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WM = TM * 32, WN = TN * 32, BM = WGM * WM, BN = WGN * WN;
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    auto epilogue_scalar = [&](int m0, int n0) {
        const int lc = lane & 31, lr = 4 * (lane >> 5);
        constexpr int TNO = (EPI == EPI_GLU) ? TN / 2 : TN;
#pragma unroll
        for (int j = 0; j < TNO; ++j) {
            const int col = n0 + wn * (EPI == EPI_GLU ? WN / 2 : WN) + j * 32 + lc;
            if (col >= g.N) continue;
            const float bias = g.bias ? g.bias[col] : 0.0f;
            float bias_g = 0.0f;
            if constexpr (EPI == EPI_GLU) bias_g = g.bias ? g.bias[g.N + col] : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + lr;
                    if (row >= g.M) continue;
                    float v = acc[i][j][r];
                    if (g.bias) v = v + bias;
                    if constexpr (EPI == EPI_RELU) {
                        v = v > 0.0f ? v : 0.0f;
                    } else if constexpr (EPI == EPI_SILU) {
                        v = gf.fast_act ? fast_siluf(v) : dsiluf(v);
                    } else if constexpr (EPI == EPI_RESID) {
                        const float y = v * g.alpha;
                        v = g.resid[(int64_t)row * g.ldr + col] + y;
                    } else if constexpr (EPI == EPI_GLU) {
                        float gt = acc[i][j + TN / 2][r];
                        if (g.bias) gt = gt + bias_g;
                        v = v * (gf.fast_act ? fast_sigmoidf(gt) : dsigmoidf(gt));
                    }
                    if (g.remap_rows) g.out[(int64_t)(row / g.remap_rows) * g.remap_gs + (int64_t)(row % g.remap_rows) * g.remap_rs + (int64_t)col * g.remap_cs] = v;
                    else g.out[(int64_t)row * g.ldo + (col < g.sigma_cols ? ((col & ~15) | ((col & 3) << 2) | ((col >> 2) & 3)) : col)] = v;
                }
            }
        }
    };
    // Wide epilogue (row-major outputs whose rows are 16-byte aligned): the accumulators go through LDS (the staging buffers
    // are free by now) so that every thread finishes 4 CONSECUTIVE output columns -- one 16-byte store per 4 results, a
    // wave writes whole 256-512 B row segments instead of 32 x 128 B slivers, the residual arrives as float4 too.
    auto epilogue_wide = [&](int m0, int n0) {
        constexpr int CP = BN + 4;                                  // C tile pitch (rows stay 16-byte aligned)
        // the C tile goes through the staging buffers; when it does not fit (single-buffered variant) in NPASS row bands
        constexpr size_t CAP = (size_t)CAP_FLOATS;
        constexpr int NPASS = ((size_t)BM * CP <= CAP) ? 1 : ((size_t)BM / 2 * CP <= CAP && (BM / 2) % 32 == 0) ? 2 : ((size_t)BM / 4 * CP <= CAP && (BM / 4) % 32 == 0) ? 4
                              : ((size_t)BM / 8 * CP <= CAP && (BM / 8) % 32 == 0) ? 8 : BM / 32, PR = BM / NPASS;   // (last resort: bands of one MFMA tile row)
        static_assert((size_t)PR * CP <= CAP, "C tile band must fit in the staging buffers");
        static_assert(PR % 32 == 0, "row bands are whole MFMA tiles");
        constexpr int C4 = NOUT / 4, NCH = BM * C4 / NT, RSTEP = NT / C4;   // float4 chunks per output row / per thread; row stride
        static_assert((BM * C4) % NT == 0 && NT % C4 == 0, "output tile must split evenly over the threads");
        static_assert(NPASS == 1 || PR % RSTEP == 0, "row bands must split evenly over the threads' row stride");
        // A thread owns the SAME 4 output columns in all of its NCH chunks (rows rl0, rl0 + RSTEP, ...).  Everything the epilogue
        // needs from global memory is requested FIRST -- bias once, the residual rows of all chunks -- so that the latency
        // (1-2 us while the co-resident workgroup streams its tiles) overlaps the LDS transposition instead of being paid once
        // per chunk: measured per-workgroup epilogue 15-28 us -> see profiles/r01_gemm_sweep_v5.txt.
        const int c4 = tid % C4, rl0 = tid / C4;
        const int col0 = n0 + 4 * c4;
        const bool col_ok = col0 < g.N;
        const bool sig = col0 < g.sigma_cols;
        const int blk = (4 * c4) & ~15, sa = c4 & 3;                // sigma layout: position 16b + 4a + e holds natural column 16b + 4e + a
        float bs[4] = {0.0f, 0.0f, 0.0f, 0.0f}, bg[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (g.bias && col_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bs[e] = g.bias[sig ? n0 + blk + 4 * e + sa : col0 + e];
                if constexpr (EPI == EPI_GLU) bg[e] = g.bias[g.N + col0 + e];
            }
        }
        constexpr int NRS = RS_PER_PASS ? NCH / NPASS : NCH;
        static_assert(!RS_PER_PASS || NCH % NPASS == 0, "chunks must split evenly over the passes");
        float4 rs[NRS];
        if constexpr (EPI == EPI_RESID && !RS_PER_PASS) {
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                int row = m0 + rl0 + q * RSTEP;
                row = row < g.M ? row : g.M - 1;
                rs[q] = col_ok ? *reinterpret_cast<const float4 *>(g.resid + (int64_t)row * g.ldr + col0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
        if constexpr (EPI == EPI_RESID && RS_PER_PASS) {
#pragma unroll
            for (int q = 0; q < NRS; ++q) {
                int row = m0 + rl0 + (pass * NRS + q) * RSTEP;
                row = row < g.M ? row : g.M - 1;
                rs[q] = col_ok ? *reinterpret_cast<const float4 *>(g.resid + (int64_t)row * g.ldr + col0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
        __syncthreads();                                            // every wave is done reading its last fragments / the previous band
        {
            const int lc = lane & 31, lr = 4 * (lane >> 5);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rb = wm * WM + i * 32;                    // first tile row of this accumulator block
                if (rb / PR != pass) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        smem[(rb - pass * PR + (r & 3) + 8 * (r >> 2) + lr) * CP + wn * WN + j * 32 + lc] = acc[i][j][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            if ((q * RSTEP) / PR != pass) continue;
            const int rl = rl0 + q * RSTEP;
            const int row = m0 + rl;
            const int rs_ = rl - pass * PR;                        // row inside the LDS band
            float v[4], gt[4];
            if (sig) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = smem[rs_ * CP + blk + 4 * e + sa];
            } else {
                int vc = 4 * c4;                                    // virtual column of the value inside the C tile
                if constexpr (EPI == EPI_GLU) vc = (vc / (WN / 2)) * WN + vc % (WN / 2);
                const float4 x = *reinterpret_cast<const float4 *>(smem + rs_ * CP + vc);
                v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
                if constexpr (EPI == EPI_GLU) {
                    const float4 y = *reinterpret_cast<const float4 *>(smem + rs_ * CP + vc + WN / 2);
                    gt[0] = y.x; gt[1] = y.y; gt[2] = y.z; gt[3] = y.w;
                }
            }
            float rsv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (EPI == EPI_RESID) { const float4 &rr = rs[RS_PER_PASS ? q - pass * NRS : q]; rsv[0] = rr.x; rsv[1] = rr.y; rsv[2] = rr.z; rsv[3] = rr.w; }
            if (g.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = v[e] + bs[e];
                    if constexpr (EPI == EPI_GLU) gt[e] = gt[e] + bg[e];
                }
            }
            if constexpr (EPI == EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
            } else if constexpr (EPI == EPI_SILU) {
                if (gf.fast_act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fast_siluf(v[e]);
                } else {
                    dsilu4(v);                                      // the specification's values in 22 instead of 34 operations (pk_devmath.h)
                }
            } else if constexpr (EPI == EPI_RESID) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float y = v[e] * g.alpha; v[e] = rsv[e] + y; }
            } else if constexpr (EPI == EPI_GLU) {
                if (gf.fast_act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) gt[e] = fast_sigmoidf(gt[e]);
                } else {
                    dsigmoid4(gt);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] * gt[e];
            }
            if (row < g.M && col_ok) {
                if (gf.out_bf16) {                                  // bf16 activations (gemm_bf16.hpp): 4 results = 8 bytes
                    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
                    const bf16x4_ o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *reinterpret_cast<bf16x4_ *>(reinterpret_cast<__bf16 *>(g.out) + (int64_t)row * g.ldo + col0) = o;
                } else {
                    *reinterpret_cast<float4 *>(g.out + (int64_t)row * g.ldo + col0) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        }
    };
    const bool wide = g.remap_rows == 0 && (g.ldo & 3) == 0 && (g.N & 3) == 0 && (EPI != EPI_RESID || (g.ldr & 3) == 0);   // (out_bf16 requires it: launcher)
    if (wide) epilogue_wide(m0, n0);
    else epilogue_scalar(m0, n0);
}

// LNA (round 6): the LayerNorm of the product's input rows applied WHILE THE A TILE IS STAGED -- A = the un-normalised rows, GemmArgs::ln_stats = {mean, rstd}
// per row (launch_layernorm_stats: the canonical sum64 reductions of layernorm_kernel), ln_g / ln_b = gamma / beta over K (= the row length).  Between
// the global load of a chunk and its LDS store every element goes through exactly layernorm_kernel's y = fma((x - mean) * rstd, gamma, beta): the
// values that reach the MFMAs are bit for bit the ones the separate launch would have written -- but they are never written (nor read back): a
// batch's LayerNorm becomes a statistics pass (one read of x, 8 bytes per row out) instead of a read + write of the whole tensor.
template <int WGM, int WGN, int TM, int TN, int BK, int EPI, int NBUF = 2, bool LNA = false>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_pipe_kernel(GemmArgs g, int tiles_n, int n_tiles) {
    static_assert(NBUF == 1 || NBUF == 2, "LDS staging buffers");
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WM = TM * 32, WN = TN * 32, BM = WGM * WM, BN = WGN * WN;
    constexpr int PITCH = BK + 4, BUF = (BM + BN) * PITCH, NSUB = BK / 8, C4R = BK / 4;   // C4R float4 chunks per tile row
    constexpr int A_CH = BM * C4R / NT, W_CH = BN * C4R / NT;                              // staging chunks per thread per K tile
    static_assert(BK == 32 || BK == 64, "BK");
    static_assert((BM * C4R) % NT == 0 && (BN * C4R) % NT == 0, "tile rows must split evenly over the threads");
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;      // output columns per block
    static_assert(EPI != EPI_GLU || (TN % 2 == 0), "GLU needs an even number of column tiles per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // Wave priority 3: the encoder's GEMM waves win instruction arbitration against the decode-loop waves of the previous batch that share
    // the SIMD (default priority 0, latency-tolerant: their stream has 3-7x slack).  rocprofv3 trace of round 2: a GEMM launch that overlaps
    // decode kernels ran 8-27 % longer (fc2, one workgroup per CU and a single round, the most).
    __builtin_amdgcn_s_setprio(3);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nk = g.K / BK;

    // staging chunk c -> (tile row, float4 column); rows are interleaved in groups of 8 (0,4,1,5,2,6,3,7) so the two
    // rows a 16-lane ds_write_b64 group touches sit 16 banks apart
    auto chunk_row = [](int c) { const int rr = c / C4R; return (rr & ~7) | ((rr & 1) << 2) | ((rr >> 1) & 3); };
    const float *a_src[A_CH];
    const float *w_src[W_CH];
    int a_dst[A_CH], w_dst[W_CH];
    [[maybe_unused]] float a_mean[LNA ? A_CH : 1], a_rstd[LNA ? A_CH : 1];     // LNA: statistics of the row each staging chunk belongs to
    [[maybe_unused]] const float *gam_src = nullptr, *bet_src = nullptr;      // LNA: gamma / beta of this thread's 4 k of every K tile (c % C4R is the same for all its chunks)
    static_assert(!LNA || NT % C4R == 0, "LNA: a thread's chunks share their k offset");
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int c = tid + NT * i;
        a_dst[i] = chunk_row(c) * PITCH + 2 * (c % C4R);
    }
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
        const int c = tid + NT * i;
        w_dst[i] = (BM + chunk_row(c)) * PITCH + 2 * (c % C4R);
    }
    // XCD-aware bijective remap (block b runs on XCD b % 8): XCD x gets a contiguous range of tiles.
    auto remap = [&](int b) {
        const int q = n_tiles >> 3, r = n_tiles & 7, xcd = b & 7, idx = b >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    };
    auto set_tile = [&](int t, int &m0, int &n0) {
        const int bid = remap(t);
        m0 = (bid / tiles_n) * BM;
        n0 = (bid % tiles_n) * NOUT;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int c = tid + NT * i;
            int gr = m0 + chunk_row(c);
            gr = gr < g.M ? gr : g.M - 1;
            a_src[i] = g.A + (int64_t)gr * g.lda + (c % C4R) * 4;
            if constexpr (LNA) {
                const float2 st = *reinterpret_cast<const float2 *>(g.ln_stats + 2 * (int64_t)gr);
                a_mean[i] = st.x; a_rstd[i] = st.y;
            }
        }
        if constexpr (LNA) { gam_src = g.ln_g + (tid % C4R) * 4; bet_src = g.ln_b + (tid % C4R) * 4; }
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            const int c = tid + NT * i, v = chunk_row(c);
            int wr;
            if constexpr (EPI == EPI_GLU) {
                // virtual column v -> (wave column, tile, lane column); tiles [0,TN/2) are the value half, tiles
                // [TN/2,TN) the gate half of the SAME output columns, so one lane holds both.
                constexpr int HT = TN / 2;
                const int vw = v / WN, rem = v % WN, tn = rem >> 5, cc = rem & 31;
                int col = n0 + vw * (WN / 2) + (tn % HT) * 32 + cc;
                col = col < g.N ? col : g.N - 1;
                wr = (tn / HT) * g.N + col;
            } else {
                wr = n0 + v;
                wr = wr < g.N ? wr : g.N - 1;
            }
            w_src[i] = g.W + (int64_t)wr * g.ldw + (c % C4R) * 4;
        }
    };

    float4 ra[A_CH], rw[W_CH];
    [[maybe_unused]] float4 rg4, rb4;
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) ra[i] = *reinterpret_cast<const float4 *>(a_src[i] + kt * BK);
#pragma unroll
        for (int i = 0; i < W_CH; ++i) rw[i] = *reinterpret_cast<const float4 *>(w_src[i] + kt * BK);
        if constexpr (LNA) {
            rg4 = *reinterpret_cast<const float4 *>(gam_src + kt * BK);
            rb4 = *reinterpret_cast<const float4 *>(bet_src + kt * BK);
        }
    };
    // Staging stores: {a, b} at p, {c, d} at p + BK/2 floats.  FOUR 4-byte stores, not one ds_write2_b64: on gfx950 the finer the LDS store
    // next to the fragment reads, the less it costs the loop -- 16-byte stores 98 TF, 8-byte pairs 134, 4-byte stores 141 TF of the 145 TF
    // MFMA rate on the single-buffered 128x128 kernel (profiles/r02_gemm_lds_store_width.txt).  Inline, because the compiler would pair
    // them again; the barrier that publishes them is preceded by lds_store_fence().
    auto st2 = [&](float *p, float a, float b, float c, float d) {
        // The products without an epilogue function (qkv: 756 tiles = 1.48 rounds of workgroups) keep the 8-byte pair form: there the 4-byte
        // stores shift the round structure the wrong way (115.6 vs 107.3 us in the sweep, +19 % in the engine).
        if (EPI == EPI_NONE) {
            *reinterpret_cast<float2 *>(p) = make_float2(a, b);
            *reinterpret_cast<float2 *>(p + BK / 2) = make_float2(c, d);
        } else {
            const unsigned addr = (unsigned)(size_t)p;              // low 32 bits of a flat LDS address = the LDS offset
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4\n\tds_write_b32 %0, %3 offset:%5\n\tds_write_b32 %0, %4 offset:%6"
                         ::"v"(addr), "v"(a), "v"(b), "v"(c), "v"(d), "n"(BK / 2 * 4), "n"(BK / 2 * 4 + 4) : "memory");
        }
    };
    // LNA: the A chunks in flight (the K tile the NEXT lstore will publish) are normalised in their registers -- layernorm_kernel's expression, operation
    // for operation (-ffp-contract=off: sub, mul, fma; two elements per instruction: v_pk_add / v_pk_mul / v_pk_fma_f32 are the same IEEE operations per
    // element).  Called BEFORE the barrier that precedes lstore, under the MFMAs of a sub-step: inside lstore the 24 dependent VALU operations sat in the
    // barrier-to-barrier section every wave of the workgroup has to leave before the next K tile can be read (first form of round 6: fc1 152 -> 163 us).
    auto lnorm = [&]() {
        if constexpr (LNA) {
            typedef float f2_ __attribute__((ext_vector_type(2)));
            const f2_ g01 = {rg4.x, rg4.y}, g23 = {rg4.z, rg4.w}, b01 = {rb4.x, rb4.y}, b23 = {rb4.z, rb4.w};
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const f2_ m = {a_mean[i], a_mean[i]}, r = {a_rstd[i], a_rstd[i]};
                f2_ lo = {ra[i].x, ra[i].y}, hi = {ra[i].z, ra[i].w};
                lo = __builtin_elementwise_fma((lo - m) * r, g01, b01);
                hi = __builtin_elementwise_fma((hi - m) * r, g23, b23);
                ra[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    };
    auto lstore = [&](int buf) {
        float *base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) st2(base + a_dst[i], ra[i].x, ra[i].z, ra[i].y, ra[i].w);       // k = 4c, 4c+2 | k = 4c+1, 4c+3
#pragma unroll
        for (int i = 0; i < W_CH; ++i) st2(base + w_dst[i], rw[i].x, rw[i].z, rw[i].y, rw[i].w);
    };

    gp_f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    };

    // fragment base of this lane: row (lane & 31) of the wave's sub-tile, half h = lane >> 5
    const int fa_off = (wm * WM + (lane & 31)) * PITCH + (BK / 2) * (lane >> 5);
    const int fb_off = (BM + wn * WN + (lane & 31)) * PITCH + (BK / 2) * (lane >> 5);
    float4 fa[2][TM], fb[2][TN];
    auto fragload = [&](int buf, int s, int slot) {
        const float *base = smem + buf * BUF + 4 * s;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const float4 *>(base + fa_off + i * 32 * PITCH);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const float4 *>(base + fb_off + j * 32 * PITCH);
    };
    auto mma = [&](int slot) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float a = e == 0 ? fa[slot][i].x : e == 1 ? fa[slot][i].y : e == 2 ? fa[slot][i].z : fa[slot][i].w;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float b = e == 0 ? fb[slot][j].x : e == 1 ? fb[slot][j].y : e == 2 ? fb[slot][j].z : fb[slot][j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                }
            }
        }
    };
    auto epilogue = [&](int m0, int n0) { gp_epilogue<WGM, WGN, TM, TN, EPI, NBUF * BUF, false, true>(g, acc, smem, m0, n0); };
#define GP_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef GP_LNORM_AT
#define GP_LNORM_AT (NSUB - 2)          // the sub-step whose MFMAs cover the normalisation (the last one in front of the barrier: the loads have had the longest to land)
#endif

    int m0, n0;
    set_tile(blockIdx.x, m0, n0);
    gload(0);
    lnorm();
    lstore(0);
    lds_store_fence();
    __syncthreads();
    gload(1);                      // nk >= 2 (K >= 2*BK, checked by the launcher)
    fragload(0, 0, 0);
    int cur = 0;
    zero_acc();
    if constexpr (NBUF == 1) {
        // Single staging buffer (half the LDS, two barriers per K tile: one when every wave has its last fragments in registers, one
        // when the next tile is stored; the last sub-step's MFMAs run between them).  Level with the double-buffered loop in the
        // engine (profiles/r02_bench_v2_sb.json); kept as a variant of the sweep.
        for (int kt = 0; kt < nk; ++kt) {
            const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
#pragma unroll
            for (int s = 0; s < NSUB - 1; ++s) {
                fragload(0, s + 1, (s + 1) & 1);
                GP_SB(); mma(s & 1); GP_SB();
                if (LNA && s == GP_LNORM_AT && more1) { lnorm(); GP_SB(); }   // (K tile kt + 1, requested most of a K tile ago, under this sub-step's MFMAs)
            }
            __syncthreads();
            if (more1) lstore(0);
            if (more2) gload(kt + 2);
            GP_SB(); mma((NSUB - 1) & 1); GP_SB();
            lds_store_fence();                                      // the staging stores are inline (st2)
            __syncthreads();
            if (more1) fragload(0, 0, 0);
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
#pragma unroll
            for (int s = 0; s < NSUB - 1; ++s) {
                fragload(cur, s + 1, (s + 1) & 1);
                if (LNA && s == 0 && more1) lnorm();
                if (s == NSUB - 2 && more1) lstore(cur ^ 1);
                GP_SB(); mma(s & 1); GP_SB();
            }
            lds_store_fence();
            __syncthreads();
            if (more1) fragload(cur ^ 1, 0, 0);
            if (more2) gload(kt + 2);
            GP_SB(); mma((NSUB - 1) & 1); GP_SB();
            cur ^= 1;
        }
    }
    epilogue(m0, n0);
#undef GP_SB
}

template <int WGM, int WGN, int TM, int TN, int BK, int EPI, int NBUF = 2, bool LNA = false>
static void launch_gemm_pipe(const GemmArgs &a, hipStream_t s) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + NOUT - 1) / NOUT;
    const int n_tiles = tiles_m * tiles_n;
    constexpr size_t lds = NBUF * (size_t)(BM + BN) * (BK + 4) * sizeof(float);
    if (a.fast_act || a.out_bf16) { fprintf(stderr, "parakeet_amd: internal error: bf16-mode switches on the fp32 GEMM\n"); abort(); }
    if (LNA != (a.ln_stats != nullptr)) { fprintf(stderr, "parakeet_amd: internal error: GemmArgs::ln_stats on a kernel without the folded LayerNorm (or the reverse)\n"); abort(); }
    auto kern = &gemm_pipe_kernel<WGM, WGN, TM, TN, BK, EPI, NBUF, LNA>;
    static DynLdsSlots slots;
    ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles);
}

}  // namespace pk
#endif  // PK_GEMM_PIPE_HPP
