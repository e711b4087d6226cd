// tools/ubench/gemm_bf16_exp.hpp -- INSTRUMENTED copy of parakeet.cpp_amd/csrc/kernels/gemm_bf16.hpp (BG_EXP component switches) for the
// micro-benchmarks; defines the product header's include guard so that including it first shadows the product version.
// parakeet.cpp_amd/csrc/kernels/gemm_bf16.hpp -- bf16-input / fp32-accumulate MFMA GEMM (gfx950), the precision
// BASELINE configs[2] (tdt-600m) names.
//
// out[M][N] = epi(bf16(A)[M][K] * W16[N][K]^T + bias): the activations stay fp32 in HBM and are rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32) while they are staged into LDS (as ds_write2_b64 pairs: 16-byte ds_write_b128 stores halve the kernel's rate); the weights are rounded once at upload and live in HBM as bf16.
// Products of two bf16 are exact in fp32 and the accumulator is fp32, so the only difference to the fp32 path is the
// operand rounding plus the summation order inside v_mfma_f32_32x32x16_bf16 (not a sequential chain): results are
// compared with the oracle's bf16 mode within a stated tolerance, not bit for bit (tests/test_gpu_bf16.py).
//
// Structure = gemm_pipe.hpp (register-double-buffered fragments, LDS double buffer, one barrier per K tile, its epilogue
// through LDS with 16-byte stores), with BK = 64: a tile row is 64 bf16 = 128 B (+16 B pad: the same 144-byte pitch, so
// the conflict-free ds_read_b128 analysis carries over); lane (row, h) reads the 8 consecutive k = 16s + 8h .. +7 of
// MFMA step s with one ds_read_b128 -- bf16 needs no K permutation.
#ifndef PK_GEMM_BF16_HPP
#define PK_GEMM_BF16_HPP
#include "gemm_pipe_exp.hpp"

#ifndef BG_GROUPM
#define BG_GROUPM 8                 // tile rows per group of the grouped tile order (1 = row-major)
#endif
#ifndef BG_EXP
#define BG_EXP 0                    // micro-benchmark experiments only (tools/ubench, results wrong, timing only): main loop without
#endif                              // 16 = LDS stores, 32 = global loads, 64 = fragment reads; 512 = staging stores as ds_write_b128 instead of ds_write2_b64 pairs, 1024 = as two separate ds_write_b64, 2048 = as four ds_write_b32

namespace pk {

typedef float bg_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bg_bf16x8 __attribute__((ext_vector_type(8)));

template <int WGM, int WGN, int TM, int TN, int EPI, bool A16 = false>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_bf16_kernel(GemmArgs g, int tiles_n, int n_tiles) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WM = TM * 32, WN = TN * 32, BM = WGM * WM, BN = WGN * WN;
    constexpr int BK = 64, PITCH = BK + 8, BUF = (BM + BN) * PITCH, NSUB = BK / 16;     // in bf16 elements
    constexpr int A_CH = BM * 8 / NT, W_CH = BN * 8 / NT;                                // 8-element chunks per thread per K tile
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile rows must split evenly over the threads");
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
    static_assert(EPI != EPI_GLU || (TN % 2 == 0), "GLU needs an even number of column tiles per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16 *smem = reinterpret_cast<__bf16 *>(smem_raw);
    float *smem_f = reinterpret_cast<float *>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nk = g.K / BK;
    const __bf16 *W16 = reinterpret_cast<const __bf16 *>(g.W);

    int bid = blockIdx.x;
    {   // XCD-aware bijective remap (block b runs on XCD b % 8): XCD x gets a contiguous range of tiles
        const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // Grouped order: BG_GROUPM tile rows down before the next tile column, so that the ~64 tiles an XCD works on at one time form a squarer
    // block of the output.  At bf16 rates this kernel lives on its L2 hit rate: in plain row-major order the tdt-600m fc1 product pulled
    // 711 MB per launch through the L2 miss path (21x its operands, 4.8 TB/s: profiles/r02_pmc_hbm_600m_bf16.json) -- every pair of tile
    // rows streamed the whole 8.4 MB weight matrix again.
    int m0, n0;
    {
        const int tiles_m = n_tiles / tiles_n, per_group = BG_GROUPM * tiles_n;
        const int grp = bid / per_group, first_m = grp * BG_GROUPM;
        const int gsz = (tiles_m - first_m) < BG_GROUPM ? (tiles_m - first_m) : BG_GROUPM;
        const int in = bid - grp * per_group;
        m0 = (first_m + in % gsz) * BM;
        n0 = (in / gsz) * NOUT;
    }

    const float *a_src[A_CH];                                       // A16: bf16 rows (g.A reinterpreted, lda in elements)
    const __bf16 *w_src[W_CH];
    const __bf16 *A16p = reinterpret_cast<const __bf16 *>(g.A);
    int a_dst[A_CH], w_dst[W_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int c = tid + NT * i, row = c >> 3, c8 = c & 7;
        int gr = m0 + row;
        gr = gr < g.M ? gr : g.M - 1;
        a_src[i] = A16 ? reinterpret_cast<const float *>(A16p + (int64_t)gr * g.lda + c8 * 8) : g.A + (int64_t)gr * g.lda + c8 * 8;
        a_dst[i] = row * PITCH + c8 * 8;
    }
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
        const int c = tid + NT * i, v = c >> 3, c8 = c & 7;
        int wr;
        if constexpr (EPI == EPI_GLU) {
            constexpr int HT = TN / 2;     // tiles [0,HT) = value half, [HT,TN) = gate half of the SAME output columns
            const int vw = v / WN, rem = v % WN, tn = rem >> 5, cc = rem & 31;
            int col = n0 + vw * (WN / 2) + (tn % HT) * 32 + cc;
            col = col < g.N ? col : g.N - 1;
            wr = (tn / HT) * g.N + col;
        } else {
            wr = n0 + v;
            wr = wr < g.N ? wr : g.N - 1;
        }
        w_src[i] = W16 + (int64_t)wr * g.ldw + c8 * 8;
        w_dst[i] = (BM + v) * PITCH + c8 * 8;
    }

    float4 ra[A_CH][2];
    uint4 rw[W_CH];
    int opaque0 = 0;
    if (BG_EXP & 1024) asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            if constexpr (A16) {
                ra[i][0] = *reinterpret_cast<const float4 *>(a_src[i] + kt * (BK / 2));      // 8 bf16 = 16 bytes, already rounded by the producer
            } else {
                ra[i][0] = *reinterpret_cast<const float4 *>(a_src[i] + kt * BK);
                ra[i][1] = *reinterpret_cast<const float4 *>(a_src[i] + kt * BK + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < W_CH; ++i) rw[i] = *reinterpret_cast<const uint4 *>(w_src[i] + kt * BK);
    };
    auto lstore = [&](int buf) {
        __bf16 *base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            if constexpr (A16) {
                if (BG_EXP & 2048) lds_store16_b32(base + a_dst[i], ra[i][0]);
                else lds_store16(base + a_dst[i], ra[i][0]);
                continue;
            }
            bg_bf16x8 v;
            v[0] = (__bf16)ra[i][0].x; v[1] = (__bf16)ra[i][0].y; v[2] = (__bf16)ra[i][0].z; v[3] = (__bf16)ra[i][0].w;
            v[4] = (__bf16)ra[i][1].x; v[5] = (__bf16)ra[i][1].y; v[6] = (__bf16)ra[i][1].z; v[7] = (__bf16)ra[i][1].w;
            if (BG_EXP & 1024) {                                   // two separate 8-byte stores the compiler cannot merge (opaque zero offset)
                const float4 q = *reinterpret_cast<const float4 *>(&v);
                *reinterpret_cast<float2 *>(base + a_dst[i]) = make_float2(q.x, q.y);
                *reinterpret_cast<float2 *>(base + a_dst[i] + 4 + opaque0) = make_float2(q.z, q.w);
            } else if (BG_EXP & 2048) {
                lds_store16_b32(base + a_dst[i], *reinterpret_cast<const float4 *>(&v));
            } else if (!(BG_EXP & 512)) {
                lds_store16(base + a_dst[i], *reinterpret_cast<const float4 *>(&v));
            } else {
                *reinterpret_cast<bg_bf16x8 *>(base + a_dst[i]) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            if (BG_EXP & 1024) {
                const float4 q = *reinterpret_cast<const float4 *>(&rw[i]);
                *reinterpret_cast<float2 *>(base + w_dst[i]) = make_float2(q.x, q.y);
                *reinterpret_cast<float2 *>(base + w_dst[i] + 4 + opaque0) = make_float2(q.z, q.w);
            } else if (BG_EXP & 2048) {
                lds_store16_b32(base + w_dst[i], *reinterpret_cast<const float4 *>(&rw[i]));
            } else if (!(BG_EXP & 512)) {
                lds_store16(base + w_dst[i], *reinterpret_cast<const float4 *>(&rw[i]));
            } else {
                *reinterpret_cast<uint4 *>(base + w_dst[i]) = rw[i];
            }
        }
    };

    bg_f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int fa_off = (wm * WM + (lane & 31)) * PITCH + 8 * (lane >> 5);
    const int fb_off = (BM + wn * WN + (lane & 31)) * PITCH + 8 * (lane >> 5);
    bg_bf16x8 fa[2][TM], fb[2][TN];
    auto fragload = [&](int buf, int s, int slot) {
        const __bf16 *base = smem + buf * BUF + 16 * s;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const bg_bf16x8 *>(base + fa_off + i * 32 * PITCH);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const bg_bf16x8 *>(base + fb_off + j * 32 * PITCH);
    };
    auto mma = [&](int slot) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot][i], fb[slot][j], acc[i][j], 0, 0, 0);
    };
#define BG_SB() __builtin_amdgcn_sched_barrier(0)
    gload(0);
    lstore(0);
    lds_store_fence();
    __syncthreads();
    if (nk > 1) gload(1);
    fragload(0, 0, 0);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
#pragma unroll
        for (int s = 0; s < NSUB - 1; ++s) {
            if (!(BG_EXP & 64)) fragload(cur, s + 1, (s + 1) & 1);
            if (!(BG_EXP & 16) && s == NSUB - 2 && more1) lstore(cur ^ 1);
            BG_SB(); mma(s & 1); BG_SB();
        }
        lds_store_fence();                                          // the staging stores are inline ds_write2_b64 (lds_store16)
        __syncthreads();
        if (!(BG_EXP & 64) && more1) fragload(cur ^ 1, 0, 0);
        if (!(BG_EXP & 32) && more2) gload(kt + 2);
        BG_SB(); mma((NSUB - 1) & 1); BG_SB();
        cur ^= 1;
    }
#undef BG_SB

    // epilogue shared with the fp32 kernels (gemm_pipe.hpp): accumulators -> LDS -> 4 consecutive columns per thread, in row bands when the
    // C tile is larger than the staging buffers
    gp_epilogue<WGM, WGN, TM, TN, EPI, BUF>(g, acc, smem_f, m0, n0);     // 2 buffers x BUF bf16 = BUF floats
}

template <int WGM, int WGN, int TM, int TN, int EPI, bool A16 = false>
static void launch_gemm_bf16_t(const GemmArgs &a, hipStream_t s) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + NOUT - 1) / NOUT;
    const int n_tiles = tiles_m * tiles_n;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * (64 + 8) * 2;
    auto kern = &gemm_bf16_kernel<WGM, WGN, TM, TN, EPI, A16>;
    static DynLdsSlots slots;
    ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles);
}

}  // namespace pk
#endif  // PK_GEMM_BF16_HPP
