// tools/ubench/gemm_pd.hpp -- EXPERIMENT (round 2, not used by the library: bit-equal, but 3-4 % slower than the one-tile-per-workgroup
// kernel on fc1, profiles/r02_gemm_pd.txt).  Persistent fp32 MFMA GEMM: the load stream runs across tile boundaries, the epilogue stores
// straight from the accumulators (gfx950).
//
// Same arithmetic, tiles and single-buffered K loop as gemm_pipe_kernel<..., NBUF = 1> (gemm_pipe.hpp): natural-k fma chains, bias / activation
// applied to the finished sums, bit-identical outputs.  What changes is what happens BETWEEN the tiles of a workgroup.  The timeline of the
// one-tile-per-workgroup kernel on the ffn fc1 product (tools/ubench/gemm_sweep trace, round 2: profiles/r02_gemm_wg_timeline.txt) shows the
// MFMA pipe of a CU idle or half-used for ~25 of 150 us: both co-resident workgroups sit in their epilogues (accumulators -> LDS -> SiLU ->
// stores with two barriers, 6-17 us) around the same time, the replacement workgroups then need a dispatch + a prologue (global -> LDS round
// trip, ~3 us), and the last round drains one workgroup per CU.  Here a workgroup owns a static list of tiles (t = b, b + G, ...) and
//  * the global loads of the NEXT tile's first two K tiles are issued inside the last two iterations of the current one, its first LDS tile
//    and fragments are in place when the K loop ends (the load stream never stops at a tile boundary: no second prologue, no dispatch gap);
//  * the epilogue needs no LDS and no barrier: in the 32x32 MFMA C layout one register of a half-wave is 32 consecutive columns of one row --
//    a full 128-byte line -- so bias, activation and the store run per register, while the co-resident workgroup's MFMAs keep the pipe busy
//    (a version that parked the sums in a second register set and finished them during the next K loop needed 157+ VGPRs: one workgroup
//    per CU, not built further).
// Needs a row-major output (optionally with the sigma column layout) and EPI in {NONE, RELU, SILU, RESID}.
#pragma once
#include "../../parakeet.cpp_amd/csrc/kernels/gemm_pipe.hpp"

namespace pk {

template <int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_pd_kernel(GemmArgs g, int tiles_n, int n_tiles) {
    constexpr int WGM = 4, WGN = 2, TM = 1, TN = 2, BK = 32;
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WM = TM * 32, WN = TN * 32, BM = WGM * WM, BN = WGN * WN;
    constexpr int PITCH = BK + 4, BUF = (BM + BN) * PITCH, NSUB = BK / 8, C4R = BK / 4;
    constexpr int A_CH = BM * C4R / NT, W_CH = BN * C4R / NT;
    static_assert(EPI == EPI_NONE || EPI == EPI_RELU || EPI == EPI_SILU || EPI == EPI_RESID, "epilogues of the persistent kernel");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nk = g.K / BK, G = gridDim.x;

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
    auto remap = [&](int b) {                      // XCD-aware bijection of gemm_pipe_kernel (block b runs on XCD b % 8; G % 8 == 0 keeps it there)
        const int q = n_tiles >> 3, r = n_tiles & 7, xcd = b & 7, idx = b >> 3;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    };
    auto tile_origin = [&](int t, int &m0, int &n0) {
        const int bid = remap(t);
        m0 = (bid / tiles_n) * BM;
        n0 = (bid % tiles_n) * BN;
    };
    auto set_ptrs = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int c = tid + NT * i;
            int gr = m0 + chunk_row(c);
            gr = gr < g.M ? gr : g.M - 1;
            a_src[i] = g.A + (int64_t)gr * g.lda + (c % C4R) * 4;
        }
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            const int c = tid + NT * i;
            int wr = n0 + chunk_row(c);
            wr = wr < g.N ? wr : g.N - 1;
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
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            *reinterpret_cast<float2 *>(smem + a_dst[i]) = make_float2(ra[i].x, ra[i].z);
            *reinterpret_cast<float2 *>(smem + a_dst[i] + BK / 2) = make_float2(ra[i].y, ra[i].w);
        }
#pragma unroll
        for (int i = 0; i < W_CH; ++i) {
            *reinterpret_cast<float2 *>(smem + w_dst[i]) = make_float2(rw[i].x, rw[i].z);
            *reinterpret_cast<float2 *>(smem + w_dst[i] + BK / 2) = make_float2(rw[i].y, rw[i].w);
        }
    };
    gp_f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.0f;
    };
    const int fa_off = (wm * WM + (lane & 31)) * PITCH + (BK / 2) * (lane >> 5);
    const int fb_off = (BM + wn * WN + (lane & 31)) * PITCH + (BK / 2) * (lane >> 5);
    float4 fa[2], fb[2][TN];
    auto fragload = [&](int s, int slot) {
        const float *base = smem + 4 * s;
        fa[slot] = *reinterpret_cast<const float4 *>(base + fa_off);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j] = *reinterpret_cast<const float4 *>(base + fb_off + j * 32 * PITCH);
    };
    auto mma = [&](int slot) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = e == 0 ? fa[slot].x : e == 1 ? fa[slot].y : e == 2 ? fa[slot].z : fa[slot].w;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float b = e == 0 ? fb[slot][j].x : e == 1 ? fb[slot][j].y : e == 2 ? fb[slot][j].z : fb[slot][j].w;
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0][j], 0, 0, 0);
            }
        }
    };
#define PD_SB() __builtin_amdgcn_sched_barrier(0)

    // C layout of the 32x32 MFMA: register r of block j of this lane = row m0 + lrow + (r & 3) + 8 * (r >> 2), column n0 + lcol + 32 j
    const int lrow = wm * WM + 4 * (lane >> 5), lcol = wn * WN + (lane & 31);
    auto finish = [&](int m0, int n0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + lcol + 32 * j;
            const bool col_ok = col < g.N;
            const float bias = (g.bias && col_ok) ? g.bias[col] : 0.0f;
            const int64_t ocol = col < g.sigma_cols ? ((col & ~15) | ((col & 3) << 2) | ((col >> 2) & 3)) : col;
            float rs[16];
            if constexpr (EPI == EPI_RESID) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = m0 + lrow + (r & 3) + 8 * (r >> 2);
                    row = row < g.M ? row : g.M - 1;
                    rs[r] = col_ok ? g.resid[(int64_t)row * g.ldr + col] : 0.0f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + lrow + (r & 3) + 8 * (r >> 2);
                float v = acc[0][j][r];
                if (g.bias) v = v + bias;
                if constexpr (EPI == EPI_RELU) v = v > 0.0f ? v : 0.0f;
                else if constexpr (EPI == EPI_SILU) v = dsiluf(v);
                else if constexpr (EPI == EPI_RESID) { const float y = v * g.alpha; v = rs[r] + y; }
                if (row < g.M && col_ok) g.out[(int64_t)row * g.ldo + ocol] = v;
            }
        }
    };

    int t = blockIdx.x, m0, n0;
    tile_origin(t, m0, n0);
    set_ptrs(m0, n0);
    gload(0);
    lstore();
    __syncthreads();
    gload(1);
    fragload(0, 0);
    zero_acc();
    for (;;) {
        const int tn = t + G;
        const bool more_tiles = tn < n_tiles;
        int mn0 = 0, nn0 = 0;
        if (more_tiles) tile_origin(tn, mn0, nn0);
        for (int kt = 0; kt < nk; ++kt) {
            const bool more1 = kt + 1 < nk || more_tiles, more2 = kt + 2 < nk || more_tiles;
#pragma unroll
            for (int s = 0; s < NSUB - 1; ++s) {
                fragload(s + 1, (s + 1) & 1);
                PD_SB(); mma(s & 1); PD_SB();
            }
            __syncthreads();
            if (more1) lstore();
            if (more2) {
                int kk = kt + 2;
                if (kk >= nk) {                      // the stream continues with the next tile of this workgroup
                    if (kk == nk) set_ptrs(mn0, nn0);
                    kk -= nk;
                }
                gload(kk);
            }
            PD_SB(); mma((NSUB - 1) & 1); PD_SB();
            __syncthreads();
            if (more1) fragload(0, 0);
        }
        if (!more_tiles) break;
        finish(m0, n0);                              // tile boundary: results out, next K loop (already staged) starts
        zero_acc();
        t = tn; m0 = mn0; n0 = nn0;
    }
#undef PD_SB
    finish(m0, n0);
}

// true when the product went out on the persistent kernel
template <int EPI>
static bool launch_gemm_pd(const GemmArgs &a, hipStream_t s) {
    constexpr int BM = 128, BN = 128, BK = 32;
    if (a.remap_rows != 0 || a.K % BK != 0 || a.K / BK < 2) return false;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int n_tiles = tiles_m * tiles_n;
    static int slots = 0;                              // two workgroups per CU
    if (!slots) {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        slots = 2 * cus;
    }
    if (n_tiles <= slots) return false;                // nothing to defer: every workgroup has one tile
    // equal shares: ceil(n_tiles / rounds) workgroups, a multiple of 8 (XCD-aware tile order)
    const int rounds = (n_tiles + slots - 1) / slots;
    int G = (n_tiles + rounds - 1) / rounds;
    G = (G + 7) & ~7;
    if (G > slots) G = slots;
    constexpr size_t lds = (size_t)(BM + BN) * (BK + 4) * sizeof(float);
    auto kern = &gemm_pd_kernel<EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(G), dim3(512), lds, s, a, tiles_n, n_tiles);
    return true;
}

}  // namespace pk
