// tools/ubench/gemm_dma.hpp -- EXPERIMENT (not part of the library): fp32 MFMA GEMM with DIRECT-TO-LDS staging (gfx950
// global_load_lds_dwordx4).  Bit-identical to the product kernels, measured slower (profiles/r02_gemm_sweep_dma.txt): kept so the
// measurement can be repeated.
//
// out[M][N] = epi(A[M][K] * W[N][K]^T + bias), natural-k fma chains: the same arithmetic, accumulation order and epilogues as
// gemm_pipe.hpp (bit-identical results), with a different way of feeding the MFMAs.  Round-2 ablations of the register-staged
// kernel (profiles/r02_gemm_mainloop_ablation.txt) showed its main loop running at the MFMA peak (148 TF) as soon as the
// global -> VGPR -> ds_write staging is taken out, and at 119-124 TF with it; the same tile fed by the LDS-DMA path reached 132 TF.
//  * Staging: every lane issues `global_load_lds_dwordx4`: 16 bytes (4 consecutive k of one tile row) go from global memory
//    straight into LDS -- no staging VGPRs, no ds_write instructions, no v_mov shuffles.  One wave instruction fills one 1 KB
//    block = 8 tile rows x 32 k.  The hardware places lane q's 16 bytes at (M0 base + 16 q), so WHICH (row, k chunk) lands in
//    slot q is chosen through the global address lane q reads: slot(rl, c) = 8 rl + (c ^ rl ^ (block & 1)) -- an XOR swizzle that
//    spreads the 32 rows of an MFMA operand over 16 of the 64 banks instead of 2.
//  * Fragments: the DMA cannot permute inside 16 bytes, so k stays in natural order in LDS and lane (row, h) fetches its operands
//    of two MFMA steps -- k = 4c + h and 4c + 2 + h -- with one ds_read2_b32 (two dwords, 8 bytes apart).  Step s of the chain
//    still consumes k = 2s (lanes 0-31) and 2s+1 (lanes 32-63): the natural-k order of the numerics contract.
//  * Pipeline: two LDS buffers; the DMA of K tile kt+2 is issued right after the barrier that releases the buffer of tile kt and
//    has a whole K tile of MFMAs to land (s_waitcnt vmcnt(0) + the next barrier publish it).
#pragma once
#include "../../parakeet.cpp_amd/csrc/kernels/gemm_pipe.hpp"

namespace pk {

template <int WGM, int WGN, int TM, int TN, int EPI>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_dma_kernel(GemmArgs g, int tiles_n, int n_tiles) {
    constexpr int BK = 32, NSUB = BK / 8;
    constexpr int NT = 64 * WGM * WGN, NW = WGM * WGN;
    constexpr int WM = TM * 32, WN = TN * 32, BM = WGM * WM, BN = WGN * WN;
    constexpr int BUF = (BM + BN) * BK;                       // floats per staging buffer (no padding: the swizzle spreads the banks)
    constexpr int NBLK = (BM + BN) / 8, NBPW = NBLK / NW;     // 1 KB blocks of 8 rows per K tile / per wave
    static_assert(NBLK % NW == 0, "blocks must split evenly over the waves");
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
    static_assert(EPI != EPI_GLU || (TN % 2 == 0), "GLU needs an even number of column tiles per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WGN, wn = wave % WGN;
    const int nk = g.K / BK;

    // XCD-aware bijective remap (block b runs on XCD b % 8): XCD x gets a contiguous range of tiles.
    int bid;
    {
        const int b = blockIdx.x, q = n_tiles >> 3, r = n_tiles & 7, xcd = b & 7, idx = b >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * NOUT;

    // DMA sources: wave w fills blocks w, w + NW, ...; lane q of block b holds (row 8b + q/8, k chunk (q%8) ^ (q/8) ^ (b&1)).
    // Rows 0..BM-1 of the stacked tile are A rows, BM.. are W rows.
    const float *src[NBPW];
#pragma unroll
    for (int i = 0; i < NBPW; ++i) {
        const int b = wv + NW * i;
        const int rl = lane >> 3, c = (lane & 7) ^ rl ^ (b & 1);
        const int row = 8 * b + rl;
        if (row < BM) {
            int gr = m0 + row;
            gr = gr < g.M ? gr : g.M - 1;
            src[i] = g.A + (int64_t)gr * g.lda + 4 * c;
        } else {
            const int v = row - BM;
            int wr;
            if constexpr (EPI == EPI_GLU) {
                // virtual column v -> (wave column, tile, lane column); tiles [0,TN/2) are the value half, tiles [TN/2,TN) the gate
                // half of the SAME output columns, so one lane holds both (as in gemm_pipe.hpp)
                constexpr int HT = TN / 2;
                const int vw = v / WN, rem = v % WN, tn = rem >> 5, cc = rem & 31;
                int col = n0 + vw * (WN / 2) + (tn % HT) * 32 + cc;
                col = col < g.N ? col : g.N - 1;
                wr = (tn / HT) * g.N + col;
            } else {
                wr = n0 + v;
                wr = wr < g.N ? wr : g.N - 1;
            }
            src[i] = g.W + (int64_t)wr * g.ldw + 4 * c;
        }
    }
    auto dma = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < NBPW; ++i) {
            float *dst = smem + buf * BUF + (wv + NW * i) * 256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };

    gp_f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment addresses: operand tile row r = base + (lane & 31); float offset of (row r, k chunk c) = 256 (r >> 3) + 32 (r & 7) +
    // 4 (c ^ x),  x = (r & 7) ^ ((r >> 3) & 1);  this lane reads dwords h and h + 2 of the chunk (h = lane >> 5)
    const int h = lane >> 5;
    int fa_base[TM], fa_x[TM], fb_base[TN], fb_x[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + (lane & 31);
        fa_base[i] = 256 * (r >> 3) + 32 * (r & 7) + h;
        fa_x[i] = 4 * ((r & 7) ^ ((r >> 3) & 1));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = BM + wn * WN + j * 32 + (lane & 31);
        fb_base[j] = 256 * (r >> 3) + 32 * (r & 7) + h;
        fb_x[j] = 4 * ((r & 7) ^ ((r >> 3) & 1));
    }
    float4 fa[2][TM], fb[2][TN];
    auto fragload = [&](int buf, int s, int slot) {            // sub-step s = k chunks 2s and 2s+1 = MFMA steps 4s .. 4s+3
        const float *base = smem + buf * BUF;
#ifdef GD_B128_PROXY   // TIMING PROXY (round 6; wrong operands): what the loop would cost if BOTH operands lay in global memory in a k order whose 16-byte chunks
        // are a lane's four consecutive MFMA steps (blocks of 8 k stored even-k-first) -- one ds_read_b128 per fragment instead of four 4-byte reads
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[slot][i] = *reinterpret_cast<const float4 *>(base + (fa_base[i] - h) + ((8 * s + 4 * h) ^ fa_x[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const float4 *>(base + (fb_base[j] - h) + ((8 * s + 4 * h) ^ fb_x[j]));
        return;
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float *p0 = base + fa_base[i] + ((8 * s) ^ fa_x[i]);
            const float *p1 = base + fa_base[i] + ((8 * s + 4) ^ fa_x[i]);
            fa[slot][i] = make_float4(p0[0], p0[2], p1[0], p1[2]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float *p0 = base + fb_base[j] + ((8 * s) ^ fb_x[j]);
            const float *p1 = base + fb_base[j] + ((8 * s + 4) ^ fb_x[j]);
            fb[slot][j] = make_float4(p0[0], p0[2], p1[0], p1[2]);
        }
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
#define GD_SB() __builtin_amdgcn_sched_barrier(0)
#define GD_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)         /* vmcnt(0): this wave's LDS-DMA loads have landed */

    dma(0, 0);
    if (nk > 1) dma(1, 1);
    GD_WAIT_VM0();
    __syncthreads();
    fragload(0, 0, 0);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
#pragma unroll
        for (int s = 0; s < NSUB - 1; ++s) {
            fragload(cur, s + 1, (s + 1) & 1);
            GD_SB(); mma(s & 1); GD_SB();
        }
        GD_WAIT_VM0();                     // K tile kt+1 (issued a whole tile ago) is in LDS
        __syncthreads();                   // ... for every wave; and every wave has its last fragments of tile kt: buffer `cur` is free
        if (more1) fragload(cur ^ 1, 0, 0);
        if (more2) dma(kt + 2, cur);
        GD_SB(); mma((NSUB - 1) & 1); GD_SB();
        cur ^= 1;
    }
#undef GD_SB
#undef GD_WAIT_VM0
    gp_epilogue<WGM, WGN, TM, TN, EPI, 2 * BUF>(g, acc, smem, m0, n0);
}

template <int WGM, int WGN, int TM, int TN, int EPI>
static void launch_gemm_dma(const GemmArgs &a, hipStream_t s) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + NOUT - 1) / NOUT;
    const int n_tiles = tiles_m * tiles_n;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * 32 * sizeof(float);
    auto kern = &gemm_dma_kernel<WGM, WGN, TM, TN, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles);
}

}  // namespace pk
