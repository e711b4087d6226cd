// tools/ubench/gemm_pipe_exp.hpp -- the INSTRUMENTED copy of parakeet.cpp_amd/csrc/kernels/gemm_pipe.hpp used by the micro-benchmarks only
// (component switches GP_EXP, clock stamps GP_CLOCKPROBE, late-start GP_DELAY_TICKS).  It defines the product header's include guard, so a
// benchmark that includes it BEFORE kernels/gemm.hip gets this version; the product header carries none of this scaffolding.
// Keep the arithmetic in step with the product header (tools/ubench/gemm_sweep checks both against the same reference).
//
// software-pipelined fp32 MFMA GEMM (gfx950).
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
#include "../../parakeet.cpp_amd/csrc/pk_devmath.h"
#include "../../parakeet.cpp_amd/csrc/kernels/kernels.hpp"

namespace pk {

typedef float gp_f32x16 __attribute__((ext_vector_type(16)));

#ifndef GP_EXP
#define GP_EXP 0                    // micro-benchmark experiments only (tools/ubench): 1 = epilogue without global stores,
#endif                              // 2 = epilogue without activation math; main loop without 8 = barrier, 16 = LDS stores,
                                    // 32 = global loads, 64 = fragment reads, 128 = MFMAs; 1024 = staging stores as ds_write2_b64 pairs instead of four ds_write_b32
#ifdef GP_CLOCKPROBE
__device__ long long gp_clk[4];     // micro-benchmark builds only: shader / wall clock deltas of block 0
__device__ long long *gp_trace;     // micro-benchmark builds only: [n_blocks][8] wall-clock stamps of wave 0 + hw id
#define GP_STAMP(i) do { if (gp_trace && tid == 0) gp_trace[(long long)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
__device__ long long *gp_tr2;       // micro-benchmark builds only: [n_blocks][16 waves][GP_TR2_IT][2] shader-clock stamps around the K-loop barrier
#define GP_TR2_IT 24
#else
#define GP_STAMP(i) do { } while (0)
#endif

// Epilogue shared by the GEMM kernels of this directory (bias, ReLU, SiLU, residual + alpha*y, GLU, sigma column layout) on the
// accumulators of a WGM x WGN grid of waves, each holding TM x TN 32x32 tiles.  `smem` is the kernel's staging memory (free by now),
// CAP its size in floats: the wide path turns the C tile row-major through it (in row bands when it does not fit).
// (RS_PER_PASS of the product header is accepted and ignored: this copy requests all residual rows up front)
template <int WGM, int WGN, int TM, int TN, int EPI, int CAP_FLOATS, bool RS_PER_PASS = false>
__device__ __forceinline__ void gp_epilogue(const GemmArgs &g, gp_f32x16 (&acc)[TM][TN], float *smem, int m0, int n0) {
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
                        v = g.fast_act ? fast_siluf(v) : dsiluf(v);
                    } else if constexpr (EPI == EPI_RESID) {
                        const float y = v * g.alpha;
                        v = g.resid[(int64_t)row * g.ldr + col] + y;
                    } else if constexpr (EPI == EPI_GLU) {
                        float gt = acc[i][j + TN / 2][r];
                        if (g.bias) gt = gt + bias_g;
                        v = v * (g.fast_act ? fast_sigmoidf(gt) : dsigmoidf(gt));
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
                              : ((size_t)BM / 8 * CP <= CAP && (BM / 8) % 32 == 0) ? 8 : BM / 32, PR = BM / NPASS;   // (as in the product header)
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
        float4 rs[NCH];
        if constexpr (EPI == EPI_RESID) {
#pragma unroll
            for (int q = 0; q < NCH; ++q) {
                int row = m0 + rl0 + q * RSTEP;
                row = row < g.M ? row : g.M - 1;
                rs[q] = col_ok ? *reinterpret_cast<const float4 *>(g.resid + (int64_t)row * g.ldr + col0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
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
        GP_STAMP(5);
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
            if constexpr (EPI == EPI_RESID) { rsv[0] = rs[q].x; rsv[1] = rs[q].y; rsv[2] = rs[q].z; rsv[3] = rs[q].w; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v[e];
                if (g.bias) x = x + bs[e];
                if constexpr (EPI == EPI_RELU) {
                    x = x > 0.0f ? x : 0.0f;
                } else if constexpr (EPI == EPI_SILU) {
                    if (!(GP_EXP & 2)) x = g.fast_act ? fast_siluf(x) : dsiluf(x);
                } else if constexpr (EPI == EPI_RESID) {
                    const float y = x * g.alpha;
                    x = rsv[e] + y;
                } else if constexpr (EPI == EPI_GLU) {
                    float t2 = gt[e];
                    if (g.bias) t2 = t2 + bg[e];
                    x = x * (g.fast_act ? fast_sigmoidf(t2) : dsigmoidf(t2));
                }
                v[e] = x;
            }
            if ((GP_EXP & 1) ? (row < 0) : (row < g.M && col_ok)) {
                if (g.out_bf16) {                                   // bf16 activations (gemm_bf16.hpp): 4 results = 8 bytes
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

template <int WGM, int WGN, int TM, int TN, int BK, int EPI, int NBUF = 2>
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
        }
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

    // GP_EXP & 4096 (timing ablation, results wrong): the A operand never goes through LDS -- every lane loads the 16 k of its row and K tile
    // straight from global memory (as if A were stored with the even / odd k of each block of 32 split: position 16 h + p <-> k = 2 p + h) one
    // K tile ahead; only W is staged.  Measures what halving the staging would be worth against the row-per-lane loads it needs.
    constexpr bool ADIR = (GP_EXP & 4096) != 0 && TM == 1 && BK == 32;
    float4 ra[A_CH], rw[W_CH];
    float4 ad_cur[4], ad_nxt[4];
    auto adload = [&](int kt, int m0_) {
        int r_ = m0_ + (wave / WGN) * WM + (lane & 31);
        r_ = r_ < g.M ? r_ : g.M - 1;
        const float *p_ = g.A + (int64_t)r_ * g.lda + kt * BK + 16 * (lane >> 5);
#pragma unroll
        for (int q = 0; q < 4; ++q) ad_nxt[q] = *reinterpret_cast<const float4 *>(p_ + 4 * q);
    };
    // GP_EXP & 8192 (round 6, TIMING PROXY, results wrong): the W operand never passes through registers -- it goes global -> LDS by DMA (as if the weights had
    // been copied once into a k order whose 16-byte chunks are a lane's four MFMA steps), two K tiles ahead into the free buffer of the DOUBLE-buffered loop; only
    // the A rows keep the global -> VGPR -> ds_write staging.  Half the staging loads and stores of a K tile for BN / 8 / waves DMA instructions per wave.
    constexpr bool WDMA = (GP_EXP & 8192) != 0 && NBUF == 2 && BK == 32;
    auto wdma = [&](int kt, int buf, int n0) {
        if constexpr (WDMA) {
            constexpr int NWV = NT / 64, NBLK_W = BN / 8;             // 1 KB blocks (8 rows x 32 k) of the W tile
            const int wv_ = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
            for (int i = 0; i < (NBLK_W + NWV - 1) / NWV; ++i) {
                const int b_ = wv_ + NWV * i;
                if (b_ < NBLK_W) {
                    int wr_ = n0 + 8 * b_ + (lane >> 3);
                    wr_ = wr_ < g.N ? wr_ : g.N - 1;
                    const float *src_ = g.W + (int64_t)wr_ * g.ldw + kt * BK + 4 * (lane & 7);
                    float *dst_ = smem + buf * BUF + BM * PITCH + b_ * 256;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src_, (__attribute__((address_space(3))) void *)dst_, 16, 0, 0);
                }
            }
        }
    };
    auto gload = [&](int kt) {
        if (!ADIR) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) ra[i] = *reinterpret_cast<const float4 *>(a_src[i] + kt * BK);
        }
        if constexpr (!WDMA) {
#pragma unroll
        for (int i = 0; i < W_CH; ++i) rw[i] = *reinterpret_cast<const float4 *>(w_src[i] + kt * BK);
        }
    };
    // Staging stores: {a, b} at p, {c, d} at p + BK/2 floats.  FOUR 4-byte stores, not one ds_write2_b64: on gfx950 the finer the LDS store
    // next to the fragment reads, the less it costs the loop -- 16-byte stores 98 TF, 8-byte pairs 134, 4-byte stores 141 TF of the 145 TF
    // MFMA rate on the single-buffered 128x128 kernel (profiles/r02_gemm_lds_store_width.txt).  Inline, because the compiler would pair
    // them again; the barrier that publishes them is preceded by lds_store_fence().
    auto st2 = [&](float *p, float a, float b, float c, float d) {
        // The products without an epilogue function (qkv: 756 tiles = 1.48 rounds of workgroups) keep the 8-byte pair form: there the 4-byte
        // stores shift the round structure the wrong way (115.6 vs 107.3 us in the sweep, +19 % in the engine).
        if ((GP_EXP & 1024) || EPI == EPI_NONE) {
            *reinterpret_cast<float2 *>(p) = make_float2(a, b);
            *reinterpret_cast<float2 *>(p + BK / 2) = make_float2(c, d);
        } else {
            const unsigned addr = (unsigned)(size_t)p;              // low 32 bits of a flat LDS address = the LDS offset
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4\n\tds_write_b32 %0, %3 offset:%5\n\tds_write_b32 %0, %4 offset:%6"
                         ::"v"(addr), "v"(a), "v"(b), "v"(c), "v"(d), "n"(BK / 2 * 4), "n"(BK / 2 * 4 + 4) : "memory");
        }
    };
    auto lstore = [&](int buf) {
        float *base = smem + buf * BUF;
        if (!ADIR) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) st2(base + a_dst[i], ra[i].x, ra[i].z, ra[i].y, ra[i].w);       // k = 4c, 4c+2 | k = 4c+1, 4c+3
        }
        if constexpr (!WDMA) {
#pragma unroll
        for (int i = 0; i < W_CH; ++i) st2(base + w_dst[i], rw[i].x, rw[i].z, rw[i].y, rw[i].w);
        }
    };

    gp_f32x16 acc[TM][TN];
    if (GP_EXP & 256) { float d_; asm volatile("; keep AGPRs allocatable %0" : "=a"(d_)); }
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
        if (!ADIR) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const float4 *>(base + fa_off + i * 32 * PITCH);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const float4 *>(base + fb_off + j * 32 * PITCH);
    };
    auto mma = [&](int slot, int sub = 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float4 &av = ADIR ? ad_cur[sub & 3] : fa[slot][i];
                const float a = e == 0 ? av.x : e == 1 ? av.y : e == 2 ? av.z : av.w;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float b = e == 0 ? fb[slot][j].x : e == 1 ? fb[slot][j].y : e == 2 ? fb[slot][j].z : fb[slot][j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                }
            }
        }
    };
    auto epilogue = [&](int m0, int n0) { gp_epilogue<WGM, WGN, TM, TN, EPI, NBUF * BUF>(g, acc, smem, m0, n0); };
#define GP_SB() __builtin_amdgcn_sched_barrier(0)

#ifdef GP_CLOCKPROBE
    const long long c0_ = clock64(), w0_ = wall_clock64();
#endif
    int m0, n0;
#ifdef GP_DELAY_TICKS
    // experiment: the second workgroup of every CU (blocks 256..511 of the first round) starts GP_DELAY_TICKS x 10 ns late, so that the
    // two co-resident workgroups do not reach their epilogues together
    if (blockIdx.x >= 256 && blockIdx.x < 512) {
        const long long t0_ = wall_clock64();
        while (wall_clock64() - t0_ < GP_DELAY_TICKS) __builtin_amdgcn_s_sleep(8);
    }
#endif
    GP_STAMP(0);
    set_tile(blockIdx.x, m0, n0);
    gload(0);
    if constexpr (WDMA) { wdma(0, 0, n0); wdma(1, 1, n0); }
    lstore(0);
    lds_store_fence();
    if constexpr (WDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    GP_STAMP(1);
    gload(1);                      // nk >= 2 (K >= 2*BK, checked by the launcher)
    if (ADIR) {
        adload(0, m0);
#pragma unroll
        for (int q = 0; q < 4; ++q) ad_cur[q] = ad_nxt[q];
        adload(1, m0);
    }
    fragload(0, 0, 0);
    int cur = 0;
    zero_acc();
    // GP_EXP bits 8..128 (micro-benchmark builds only; results are wrong, only the time is read) switch main-loop components off:
    // 8 = barrier, 16 = LDS stores, 32 = global loads, 64 = fragment reads, 128 = MFMAs  (profiles/r02_gemm_mainloop_ablation.txt)
    if constexpr (NBUF == 1) {
        // Single staging buffer (half the LDS, two barriers per K tile: one when every wave has its last fragments in registers, one
        // when the next tile is stored; the last sub-step's MFMAs run between them).  Level with the double-buffered loop in the
        // engine (profiles/r02_bench_v2_sb.json); kept as a variant of the sweep.
        for (int kt = 0; kt < nk; ++kt) {
            const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
#pragma unroll
            for (int s = 0; s < NSUB - 1; ++s) {
                fragload(0, s + 1, (s + 1) & 1);
                GP_SB(); mma(s & 1, s); GP_SB();
            }
            __syncthreads();
            if (more1) lstore(0);
            if (more2) gload(kt + 2);
            GP_SB(); mma((NSUB - 1) & 1, NSUB - 1); GP_SB();
            if (ADIR) {
#pragma unroll
                for (int q = 0; q < 4; ++q) ad_cur[q] = ad_nxt[q];
                if (more2) adload(kt + 2, m0);
            }
            lds_store_fence();                                      // the staging stores are inline (st2)
            __syncthreads();
            if (more1) fragload(0, 0, 0);
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
#pragma unroll
            for (int s = 0; s < NSUB - 1; ++s) {
                if (!(GP_EXP & 64)) fragload(cur, s + 1, (s + 1) & 1);
                if (!(GP_EXP & 16) && s == NSUB - 2 && more1) lstore(cur ^ 1);
                GP_SB(); if (!(GP_EXP & 128)) mma(s & 1); GP_SB();
            }
#ifdef GP_CLOCKPROBE
            unsigned long long tA_ = 0, tB_ = 0;
            if (gp_tr2) asm volatile("s_memtime %0" : "=s"(tA_));
#endif
            lds_store_fence();
            if (!(GP_EXP & 8)) __syncthreads();
#ifdef GP_CLOCKPROBE
            if (gp_tr2) {
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tB_));
                if (lane == 0 && kt < GP_TR2_IT) {
                    long long *d = gp_tr2 + (((long long)blockIdx.x * 16 + wave) * (GP_TR2_IT + 1) + kt) * 2;
                    d[0] = (long long)tA_;
                    d[1] = (long long)tB_;
                }
            }
#endif
            if (!(GP_EXP & 64) && more1) fragload(cur ^ 1, 0, 0);
            if (!(GP_EXP & 32) && more2) gload(kt + 2);
            if (WDMA && more2) wdma(kt + 2, cur, n0);               // (buffer `cur` is free: every wave holds its last fragments of tile kt)
            GP_SB(); if (!(GP_EXP & 128)) mma((NSUB - 1) & 1); GP_SB();
            if constexpr (WDMA) {   // the DMA of tile kt + 1 (issued an iteration ago) must have landed before the next barrier publishes it; what was issued THIS iteration may fly on
                constexpr int NEWER = A_CH + (BN / 8 + NT / 64 - 1) / (NT / 64);
                if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEWER) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            cur ^= 1;
        }
    }
    GP_STAMP(2);
    epilogue(m0, n0);
    GP_STAMP(3);
#undef GP_SB
#ifdef GP_CLOCKPROBE
    if (blockIdx.x == 0 && tid == 0) { gp_clk[0] = clock64() - c0_; gp_clk[1] = wall_clock64() - w0_; }
    if (gp_tr2 && lane == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        gp_tr2[(((long long)blockIdx.x * 16 + wave) * (GP_TR2_IT + 1) + GP_TR2_IT) * 2] = (long long)hw | ((long long)(xcc & 0xf) << 32);
    }
    if (gp_trace && tid == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        gp_trace[(long long)blockIdx.x * 8 + 4] = (long long)hw | ((long long)(xcc & 0xf) << 32);
    }
#endif
}

template <int WGM, int WGN, int TM, int TN, int BK, int EPI, int NBUF = 2, bool LNA = false /* round 6: the product header's folded LayerNorm -- not in this instrumented copy */>
static void launch_gemm_pipe(const GemmArgs &a, hipStream_t s) {
    if (LNA) { fprintf(stderr, "gemm_pipe_exp.hpp: no LNA instantiation\n"); abort(); }
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + NOUT - 1) / NOUT;
    const int n_tiles = tiles_m * tiles_n;
    constexpr size_t lds = NBUF * (size_t)(BM + BN) * (BK + 4) * sizeof(float);
    auto kern = &gemm_pipe_kernel<WGM, WGN, TM, TN, BK, EPI, NBUF>;
    static DynLdsSlots slots;
    ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles);
}

}  // namespace pk
#endif  // PK_GEMM_PIPE_HPP
