// tools/ubench/gemm_bf16p.hpp -- EXPERIMENT (round 2, not used by the library): the bf16-activation GEMM with a ring of PF register sets of
// global loads in flight (profiles/r02_gemm_bf16_prefetch.txt: no gain -- 128x128 pf3 = the one-tile-ahead kernel, 256x128 with 2 / 3 / 4
// tiles ahead 550-750 TF against 630-860 for 128x128).
#pragma once
#include <type_traits>
#include "../../parakeet.cpp_amd/csrc/kernels/gemm_bf16.hpp"

namespace pk {

// ---- deep-prefetch variant for bf16 activations ------------------------------------------------------------------------------------------
// Same tiles, LDS layout, fragment path and epilogue as gemm_bf16_kernel<..., A16 = true>; what changes is how far the global loads run
// ahead.  At 32 MFMA clocks per 16 k a K tile of 64 lasts 0.2-0.4 us per workgroup -- the one-tile-ahead register prefetch of the kernel
// above cannot cover an L2 round trip (~1 us under load), which is why its 256x128 / 256x256 variants lose to 128x128 although they need
// half the L1 bandwidth per flop (a 128x128 tile asks 62 B/clk/CU of the 64 the L1 delivers).  With bf16 A a K tile is 16 bytes per chunk
// for both operands, so a ring of PF register sets fits beside 64-128 accumulator registers: tile kt + PF is requested when tile kt + 1
// goes to LDS.
template <int WGM, int WGN, int TM, int TN, int EPI, int PF>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_bf16p_kernel(GemmArgs g, int tiles_n, int n_tiles) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WM = TM * 32, WN = TN * 32, BM = WGM * WM, BN = WGN * WN;
    constexpr int BK = 64, PITCH = BK + 8, BUF = (BM + BN) * PITCH, NSUB = BK / 16;     // in bf16 elements
    constexpr int A_CH = BM * 8 / NT, W_CH = BN * 8 / NT;
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile rows must split evenly over the threads");
    static_assert(EPI != EPI_GLU, "GLU stays on gemm_bf16_kernel");
    static_assert(PF >= 2 && PF <= 4, "prefetch ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16 *smem = reinterpret_cast<__bf16 *>(smem_raw);
    float *smem_f = reinterpret_cast<float *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int nk = g.K / BK;
    const __bf16 *A16p = reinterpret_cast<const __bf16 *>(g.A), *W16 = reinterpret_cast<const __bf16 *>(g.W);
    int bid = blockIdx.x;
    {
        const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const __bf16 *a_src[A_CH], *w_src[W_CH];
    int a_dst[A_CH], w_dst[W_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int c = tid + NT * i, row = c >> 3, c8 = c & 7;
        int gr = m0 + row;
        gr = gr < g.M ? gr : g.M - 1;
        a_src[i] = A16p + (int64_t)gr * g.lda + c8 * 8;
        a_dst[i] = row * PITCH + c8 * 8;
    }
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
        const int c = tid + NT * i, v = c >> 3, c8 = c & 7;
        int wr = n0 + v;
        wr = wr < g.N ? wr : g.N - 1;
        w_src[i] = W16 + (int64_t)wr * g.ldw + c8 * 8;
        w_dst[i] = (BM + v) * PITCH + c8 * 8;
    }
    float4 ra[PF][A_CH], rw[PF][W_CH];
    auto gload = [&](int kt, auto setc) {
        constexpr int set = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) ra[set][i] = *reinterpret_cast<const float4 *>(a_src[i] + kt * BK);
#pragma unroll
        for (int i = 0; i < W_CH; ++i) rw[set][i] = *reinterpret_cast<const float4 *>(w_src[i] + kt * BK);
    };
    auto lstore = [&](int buf, auto setc) {
        constexpr int set = decltype(setc)::value;
        __bf16 *base = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) lds_store16(base + a_dst[i], ra[set][i]);
#pragma unroll
        for (int i = 0; i < W_CH; ++i) lds_store16(base + w_dst[i], rw[set][i]);
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
#define BGP_SB() __builtin_amdgcn_sched_barrier(0)
    // prologue: tiles 0 .. PF-1 requested, tile 0 to LDS, its register set re-used for tile PF
    gload(0, std::integral_constant<int, 0>{});
    if (1 < nk) gload(1, std::integral_constant<int, 1>{});
    if constexpr (PF > 2) { if (2 < nk) gload(2, std::integral_constant<int, 2>{}); }
    if constexpr (PF > 3) { if (3 < nk) gload(3, std::integral_constant<int, 3>{}); }
    lstore(0, std::integral_constant<int, 0>{});
    if (PF < nk) gload(PF, std::integral_constant<int, 0>{});
    lds_store_fence();
    __syncthreads();
    fragload(0, 0, 0);
    // one K tile; tile kt + 1 sits in register set SN (static), which is refilled with tile kt + 1 + PF as soon as it has gone to LDS
    auto step = [&](int kt, auto snc) {
        const int cur = kt & 1;
        const bool more1 = kt + 1 < nk;
#pragma unroll
        for (int s = 0; s < NSUB - 1; ++s) {
            fragload(cur, s + 1, (s + 1) & 1);
            if (s == NSUB - 2 && more1) {
                lstore(cur ^ 1, snc);
                if (kt + 1 + PF < nk) gload(kt + 1 + PF, snc);
            }
            BGP_SB(); mma(s & 1); BGP_SB();
        }
        lds_store_fence();
        __syncthreads();
        if (more1) fragload(cur ^ 1, 0, 0);
        BGP_SB(); mma((NSUB - 1) & 1); BGP_SB();
    };
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
        step(kt0, std::integral_constant<int, 1 % PF>{});
        if (kt0 + 1 < nk) step(kt0 + 1, std::integral_constant<int, 2 % PF>{});
        if constexpr (PF > 2) { if (kt0 + 2 < nk) step(kt0 + 2, std::integral_constant<int, 3 % PF>{}); }
        if constexpr (PF > 3) { if (kt0 + 3 < nk) step(kt0 + 3, std::integral_constant<int, 4 % PF>{}); }
    }
#undef BGP_SB
    gp_epilogue<WGM, WGN, TM, TN, EPI, BUF>(g, acc, smem_f, m0, n0);
}

template <int WGM, int WGN, int TM, int TN, int EPI, int PF>
static void launch_gemm_bf16p_t(const GemmArgs &a, hipStream_t s) {
    constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int n_tiles = tiles_m * tiles_n;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * (64 + 8) * 2;
    auto kern = &gemm_bf16p_kernel<WGM, WGN, TM, TN, EPI, PF>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles);
}

}  // namespace pk
