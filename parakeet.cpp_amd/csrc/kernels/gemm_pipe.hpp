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
//  * PERSIST: a workgroup walks several output tiles and the pipeline runs straight through tile boundaries -- the
//    first K tiles of the next output tile are loaded, staged and fragment-read while the current tile finishes, so
//    only the epilogue itself (not the ~2 us load/stage prologue) sits between two tiles' MFMA streams.
#pragma once
#include "../pk_devmath.h"
#include "kernels.hpp"

namespace pk {

typedef float gp_f32x16 __attribute__((ext_vector_type(16)));
#ifdef GP_CLOCKPROBE
__device__ long long gp_clk[4];     // micro-benchmark builds only: shader / wall clock deltas of block 0
#endif

template <int WGM, int WGN, int TM, int TN, int BK, bool PERSIST, int EPI>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_pipe_kernel(GemmArgs g, int tiles_n, int n_tiles) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WM = TM * 32, WN = TN * 32, BM = WGM * WM, BN = WGN * WN;
    constexpr int PITCH = BK + 4, BUF = (BM + BN) * PITCH, NSUB = BK / 8, C4R = BK / 4;   // C4R float4 chunks per tile row
    constexpr int A_CH = BM * C4R / NT, W_CH = BN * C4R / NT;                              // staging chunks per thread per K tile
    static_assert(BK == 32 || BK == 64, "BK");
    static_assert((BM * C4R) % NT == 0 && (BN * C4R) % NT == 0, "tile rows must split evenly over the threads");
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;      // output columns per block
    static_assert(EPI != EPI_GLU || (TN % 2 == 0), "GLU needs an even number of column tiles per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];

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

    float4 ra[A_CH], rw[W_CH];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) ra[i] = *reinterpret_cast<const float4 *>(a_src[i] + kt * BK);
#pragma unroll
        for (int i = 0; i < W_CH; ++i) rw[i] = *reinterpret_cast<const float4 *>(w_src[i] + kt * BK);
    };
    auto lstore = [&](int buf) {
        float *base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            *reinterpret_cast<float2 *>(base + a_dst[i]) = make_float2(ra[i].x, ra[i].z);             // k = 4c, 4c+2
            *reinterpret_cast<float2 *>(base + a_dst[i] + BK / 2) = make_float2(ra[i].y, ra[i].w);    // k = 4c+1, 4c+3
        }
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            *reinterpret_cast<float2 *>(base + w_dst[i]) = make_float2(rw[i].x, rw[i].z);
            *reinterpret_cast<float2 *>(base + w_dst[i] + BK / 2) = make_float2(rw[i].y, rw[i].w);
        }
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
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    auto epilogue = [&](int m0, int n0) {
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
                        v = dsiluf(v);
                    } else if constexpr (EPI == EPI_RESID) {
                        const float y = v * g.alpha;
                        v = g.resid[(int64_t)row * g.ldr + col] + y;
                    } else if constexpr (EPI == EPI_GLU) {
                        float gt = acc[i][j + TN / 2][r];
                        if (g.bias) gt = gt + bias_g;
                        v = v * dsigmoidf(gt);
                    }
                    if (g.remap_rows) g.out[(int64_t)(row / g.remap_rows) * g.remap_gs + (int64_t)(row % g.remap_rows) * g.remap_rs + (int64_t)col * g.remap_cs] = v;
                    else g.out[(int64_t)row * g.ldo + (col < g.sigma_cols ? ((col & ~15) | ((col & 3) << 2) | ((col >> 2) & 3)) : col)] = v;
                }
            }
        }
    };
#define GP_SB() __builtin_amdgcn_sched_barrier(0)

#ifdef GP_CLOCKPROBE
    const long long c0_ = clock64(), w0_ = wall_clock64();
#endif
    int t = blockIdx.x;
    const int t_step = PERSIST ? (int)gridDim.x : n_tiles;
    int m0, n0;
    set_tile(t, m0, n0);
    gload(0);
    lstore(0);
    __syncthreads();
    gload(1);                      // nk >= 2 (K >= 2*BK, checked by the launcher)
    fragload(0, 0, 0);
    int cur = 0;
    zero_acc();
    while (true) {
        const bool has_next = PERSIST && (t + t_step < n_tiles);
        int m0n = m0, n0n = n0;
        for (int kt = 0; kt < nk; ++kt) {
            // K tiles kt+1 / kt+2 of the stream: they belong to the next output tile once they run past nk
            const bool more1 = (kt + 1 < nk) || has_next;
            const bool more2 = (kt + 2 < nk) || has_next;
#pragma unroll
            for (int s = 0; s < NSUB - 1; ++s) {
                fragload(cur, s + 1, (s + 1) & 1);
                if (s == NSUB - 2 && more1) lstore(cur ^ 1);
                GP_SB(); mma(s & 1); GP_SB();
            }
            __syncthreads();
            if (more1) fragload(cur ^ 1, 0, 0);
            if (more2) {
                if (PERSIST && kt + 2 == nk) set_tile(t + t_step, m0n, n0n);
                gload(kt + 2 < nk ? kt + 2 : kt + 2 - nk);
            }
            GP_SB(); mma((NSUB - 1) & 1); GP_SB();
            cur ^= 1;
        }
        epilogue(m0, n0);
        if (!has_next) break;
        t += t_step; m0 = m0n; n0 = n0n;
        zero_acc();
    }
#undef GP_SB
#ifdef GP_CLOCKPROBE
    if (blockIdx.x == 0 && tid == 0) { gp_clk[0] = clock64() - c0_; gp_clk[1] = wall_clock64() - w0_; }
#endif
}

// Occupancy-sized grid for the persistent form: as many workgroups as fit (LDS-limited), at most one per tile.
template <int WGM, int WGN, int TM, int TN, int BK, bool PERSIST, int EPI>
static void launch_gemm_pipe(const GemmArgs &a, hipStream_t s) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + NOUT - 1) / NOUT;
    const int n_tiles = tiles_m * tiles_n;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * (BK + 4) * sizeof(float);
    auto kern = &gemm_pipe_kernel<WGM, WGN, TM, TN, BK, PERSIST, EPI>;
    static int wg_per_cu = 0;
    if (!wg_per_cu) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void *>(kern), 64 * WGM * WGN, lds) != hipSuccess || n < 1) n = 1;
        wg_per_cu = n;
    }
    int grid = n_tiles;
    if (PERSIST) {
        const int cap = 256 * wg_per_cu;                 // 256 CUs; a multiple of 8 keeps a workgroup on one XCD's tile range
        grid = n_tiles < cap ? n_tiles : cap;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles);
}

}  // namespace pk
