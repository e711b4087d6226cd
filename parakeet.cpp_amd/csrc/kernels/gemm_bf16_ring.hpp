// parakeet.cpp_amd/csrc/kernels/gemm_bf16_ring.hpp -- bf16 MFMA GEMM as ONE CONTINUOUS operand stream per CU (round 5).
//
// out[M][N] = epi(A16[M][K] * W16[N][K]^T + bias), fp32 accumulation on v_mfma_f32_32x32x16_bf16: the arithmetic of gemm_bf16_glds.hpp (same
// products, same k order per output element: bit-identical results), restructured around what the round-5 measurement found to bound the
// K = 1024 products of tdt-600m (fc1, qkv, GLU): NOT the prologue / epilogue / re-dispatch of a tile -- a persistent launch with a register
// epilogue removes all three and gains 4 % (profiles/r05_bf16_direct_epilogue_ab.txt) -- but the K loop itself, which runs at ~0.85 PF inside the
// engine against 1.3 PF on L2-resident operands (tools/ubench/gemm_bf16_k.cpp).  With two 64-k staging buffers a K tile is requested ONE tile
// (~1 us of MFMAs) before the barrier that needs it; the XCD's 32 workgroups work on 6 MB of operands against 4 MB of L2, so part of every tile
// comes from the Infinity Cache / HBM at 1-2 us, and every workgroup waits for the slowest line at every barrier.
//   * RING of four 32-k slots (4 x 32 KB for a 256 x 256 macro tile; the same 128 KB): the DMA of K tile t+4 is issued at the barrier that
//     publishes tile t+1, i.e. THREE tiles (~1.5-3 us) ahead, 96 KB in flight per CU; s_waitcnt vmcnt(N) is COUNTED (the two younger tiles
//     stay in flight across the raw s_barrier -- cdna_hip_programming.md T3 / T4), never 0 inside a tile.
//   * the ring does not stop at the end of an output tile: the workgroup is persistent (one per CU, its XCD's contiguous range of tiles), the
//     producer cursor runs ahead into the NEXT output tile's first K tiles while the consumer finishes this one, and the epilogue is the
//     DIRECT register epilogue of gemm_bf16_glds.hpp (operands swapped in the MFMA: a lane owns an output row and four consecutive columns;
//     bias through the scalar cache; no LDS, no barrier, no vector load) -- the K stream of a CU is continuous from launch to exit.
//   * gfx950 counts loads and stores in one counter and lets them complete out of order relative to each other, so the first wait after an
//     epilogue's stores is vmcnt(0) (by then the prefetched tiles landed long ago; what it waits for is the tail of the stores), the counted
//     waits resume with the next tile of the ring.  The last tiles of the stream (nothing younger in flight) wait vmcnt(0) too.
//   * LDS image of a slot: rows of 64 B (4 chunks of 16 B), unpadded (LDS-DMA destinations are lane-linear: one wave instruction fills 16
//     rows); physical chunk p of row r holds logical chunk p ^ ((r >> 2) & 3) -- chosen through the SOURCE address of each lane -- which
//     makes the ds_read_b128 fragment reads (lane = row, logical chunk 2 s + h) conflict-free for the instruction's 16-lane groups.
// EPI_NONE / EPI_RELU / EPI_SILU / EPI_GLU, bf16 or fp32 rows out (the residual epilogue needs vector loads: gemm_bf16_glds.hpp keeps it).
#ifndef PK_GEMM_BF16_RING_HPP
#define PK_GEMM_BF16_RING_HPP
#include "gemm_bf16_glds.hpp"

namespace pk {

// PHASED (round 5, second form): the 8-phase idea of cdna_hip_programming.md section 5 on this ring.  The K loop of the forms above keeps the matrix
// pipe busy ~45 % of its clocks (tools/ubench/gemm_bf16_trace.cpp: 4500 clocks per 64-k tile against 2048 of MFMA issue) although neither the DMA's
// latency (ring), nor the LDS round trips (ASMFRAG), nor the epilogue are in the way any more: the two waves of a SIMD do the same thing at the same
// time -- both read fragments and request DMA, then both want the pipe.  Here every 32-k tile is four ITEMS per wave -- L0 (read the fragments of
// k-step 0, wait for them), M0 (8-12 MFMAs, the first half of this wave's DMA requests of tile t+3 between them), L1 (fragments of k-step 1; the
// counted wait that publishes tile t+1), M1 -- with a raw s_barrier after every item, and the second half of the waves (w + NW / 2 shares its SIMD
// with w) runs ONE ITEM BEHIND the first: while one wave of a SIMD is in an M item the other is in an L item, enforced by the barriers rather than
// hoped for.  The halves re-align around the epilogue (one extra barrier each), which both run together.  One fragment register set.
// PRIO: s_setprio 1 around the MFMA items (T5: it has something to arbitrate in this structure).
template <int WGM, int WGN, int TM, int TN, int EPI, bool STAGGER = false, bool PHASED = false, bool PRIO = false>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_bf16_ring_kernel(GemmArgs g, int tiles_n, int n_tiles) {
    constexpr int BK = 32, NR = 4;                                  // bf16 elements per slot row; ring slots
    constexpr int NW = WGM * WGN;
    constexpr int WM = TM * 32, WN = TN * 32, BM = WGM * WM, BN = WGN * WN;
    constexpr int SLOT = (BM + BN) * BK;                            // bf16 elements per slot
    // 1 KB blocks of 16 rows per K tile; per wave NBPW of them -- one fewer on the waves >= NEXTRA when the blocks do not split evenly
    // (192 x 256 tiles: 28 blocks on 8 waves), which the counted waits below account for per wave
    constexpr int NBLK = (BM + BN) / 16, NBPW = (NBLK + NW - 1) / NW, NEXTRA = NBLK % NW;
    static_assert((BM + BN) % 16 == 0, "whole 16-row blocks");
    static_assert(EPI != EPI_RESID, "the residual epilogue keeps the LDS form (gemm_bf16_glds.hpp)");
    constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
    static_assert(EPI != EPI_GLU || (TN % 2 == 0), "GLU needs an even number of column tiles per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char ring_smem_raw[];
    __bf16 *smem = reinterpret_cast<__bf16 *>(ring_smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WGN, wn = wave % WGN;
    // STAGGER: as in gemm_bf16_glds.hpp -- the second half of the waves requests its DMA pieces after the MFMA group that follows the barrier
    const bool late = STAGGER && wv >= NW / 2;
    const int nk = g.K / BK;                                        // a multiple of NR (launcher): slot of K tile kt = kt % NR in every output tile
    const __bf16 *A16 = reinterpret_cast<const __bf16 *>(g.A);
    const __bf16 *W16 = reinterpret_cast<const __bf16 *>(g.W);

    // tile order of gemm_bf16_glds.hpp: GROUPM tile rows down before the next tile column; XCD x owns a contiguous range of tiles
    auto tile_origin = [&](int bid, int &m0, int &n0) {
        constexpr int GROUPM = (BM >= 256) ? 4 : 8;
        const int tiles_m = n_tiles / tiles_n, per_group = GROUPM * tiles_n;
        const int grp = bid / per_group, first_m = grp * GROUPM;
        const int gsz = (tiles_m - first_m) < GROUPM ? (tiles_m - first_m) : GROUPM;
        const int in = bid - grp * per_group;
        m0 = (first_m + in % gsz) * BM;
        n0 = (in / gsz) * NOUT;
    };
    const int xq = n_tiles >> 3, xr = n_tiles & 7, xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    const int x_first = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq, x_count = xq + (xcd < xr ? 1 : 0);
    const int per_xcd = (int)(gridDim.x >> 3);
    if (idx >= x_count) return;
    const int my_tiles = (x_count - idx + per_xcd - 1) / per_xcd;   // output tiles this workgroup walks
    const int total = my_tiles * nk;                                // K tiles of its stream

    // DMA sources of one output tile: wave w fills blocks w, w + NW, ...; lane q of block b supplies row 16 b + q / 4, logical chunk
    // (q % 4) ^ ((row >> 2) & 3).  Rows 0 .. BM-1 of the stacked tile are A rows, BM .. BM+BN-1 are W rows.
    auto set_src = [&](int m0, int n0, const __bf16 *(&src)[NBPW]) {
#pragma unroll
        for (int i = 0; i < NBPW; ++i) {
            const int b = wv + NW * i;
            int row = 16 * b + (lane >> 2);
            row = row < BM + BN ? row : BM + BN - 1;                // (the block a short wave does not have: never requested)
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            if (row < BM) {
                int gr = m0 + row;
                gr = gr < g.M ? gr : g.M - 1;
                src[i] = A16 + (int64_t)gr * g.lda + 8 * c;
            } else {
                const int v = row - BM;
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
                src[i] = W16 + (int64_t)wr * g.ldw + 8 * c;
            }
        }
    };
    auto dma = [&](const __bf16 *const (&src)[NBPW], int kt, int slot, int i0, int i1) {
#pragma unroll
        for (int i = 0; i < NBPW; ++i) {
            if (i < i0 || i >= i1) continue;                                         // (PHASED: the pieces of one half)
            if (NEXTRA != 0 && i == NBPW - 1 && wv >= NEXTRA) continue;              // (wave-uniform: this wave has one block fewer)
            __bf16 *dst = smem + slot * SLOT + (wv + NW * i) * 512;                  // 1 KB = 512 bf16 per block; wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };

    // producer cursor: the next K tile of the stream to request (runs up to NR tiles ahead of the consumer, across output tiles)
    const __bf16 *p_src[NBPW];
    int p_loc = idx, p_kt = 0, p_m0, p_n0;
    bool p_valid = true;
    tile_origin(x_first + p_loc, p_m0, p_n0);
    set_src(p_m0, p_n0, p_src);
    auto produce = [&]() {
        if (!p_valid) return;                                       // (wave-uniform: the stream has ended)
        dma(p_src, p_kt, p_kt & (NR - 1), 0, NBPW);
        if (++p_kt == nk) {
            p_kt = 0;
            p_loc += per_xcd;
            p_valid = p_loc < x_count;
            if (p_valid) {
                tile_origin(x_first + p_loc, p_m0, p_n0);
                set_src(p_m0, p_n0, p_src);
            }
        }
    };

    // PHASED: the same cursor in two halves -- half 0 requests pieces [0, NBPW / 2) of the cursor's tile, half 1 the rest and advances
    auto produce_half = [&](int half) {
        if (!p_valid) return;
        if (half == 0) { dma(p_src, p_kt, p_kt & (NR - 1), 0, NBPW / 2); return; }
        dma(p_src, p_kt, p_kt & (NR - 1), NBPW / 2, NBPW);
        if (++p_kt == nk) {
            p_kt = 0;
            p_loc += per_xcd;
            p_valid = p_loc < x_count;
            if (p_valid) {
                tile_origin(x_first + p_loc, p_m0, p_n0);
                set_src(p_m0, p_n0, p_src);
            }
        }
    };

    bg_f32x16 acc[TM][TN];
    // fragment addresses: operand-tile row r = tile base (a multiple of 32) + (lane & 31): element offset r * 32 + ((2 s + h) ^ x) * 8 with
    // x = (r >> 2) & 3 = ((lane & 31) >> 2) & 3 -- one swizzle value per lane for every tile
    const int h = lane >> 5, fx = ((lane & 31) >> 2) & 3;
    const int fa_base = (wm * WM + (lane & 31)) * BK, fb_base = (BM + wn * WN + (lane & 31)) * BK;
    bg_bf16x8 fa[2][TM], fb[2][TN];
    auto fragload = [&](int slot, int s, int reg) {
        const __bf16 *base = smem + slot * SLOT + (((2 * s + h) ^ fx) << 3);
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[reg][i] = *reinterpret_cast<const bg_bf16x8 *>(base + fa_base + i * 32 * BK);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[reg][j] = *reinterpret_cast<const bg_bf16x8 *>(base + fb_base + j * 32 * BK);
    };
    auto mma = [&](int reg) {                                       // operands swapped (C^T): a lane owns an output row (gl_epilogue_direct)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[reg][j], fa[reg][i], acc[i][j], 0, 0, 0);
    };
#define RG_SB() __builtin_amdgcn_sched_barrier(0)
    // counted waits, by hand: N = DMA instructions of this wave that may stay in flight (NBPW per K tile)
#define RG_WAIT_N(N) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory")
    // T = K tiles of this wave's DMA that may stay in flight
#define RG_WAIT(T) do { if (NEXTRA != 0 && wv >= NEXTRA) RG_WAIT_N((T) * (NBPW - 1)); else RG_WAIT_N((T) * NBPW); } while (0)
    // raw s_barrier (a __syncthreads() would drain the LDS-DMA still in flight: vmcnt(0)); the empty asm keeps the compiler's LDS reads below it
#define RG_BARRIER() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

    if constexpr (PHASED) {
        const bool grp1 = wv >= NW / 2;                             // the half that runs one item behind
        const unsigned lds0 = (unsigned)(size_t)smem;
        auto fragload_asm = [&](int slot, int s_) {                 // hand-counted reads (gemm_bf16_glds.hpp ASMFRAG), one register set
            const unsigned e = (unsigned)(slot * SLOT + (((2 * s_ + h) ^ fx) << 3));
            const unsigned aa = lds0 + 2u * (e + (unsigned)fa_base), ab = lds0 + 2u * (e + (unsigned)fb_base);
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[0][i]) : "v"(aa), "n"(i * 32 * BK * 2));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[0][j]) : "v"(ab), "n"(j * 32 * BK * 2));
        };
        auto frag_pin = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[0][i]));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[0][j]));
        };
        // an M item: the MFMAs of one k-step with one half of this wave's DMA requests of tile t + 3 between them
        auto m_item = [&](int half) {
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
            int n = 0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][j], fa[0][i], acc[i][j], 0, 0, 0);
                    if (++n == 2) produce_half(half);
                }
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        };
        // pieces of one tile this wave requests (P) and of its first half (H0): the counted wait of L1 lets tile t + 2 and the first half of t + 3 fly
        const bool short_wave = NEXTRA != 0 && wv >= NEXTRA;
#define RG_WAIT_L1() do { if (short_wave) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NBPW - 1 + NBPW / 2) : "memory"); \
                          else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NBPW + NBPW / 2) : "memory"); } while (0)
        // prologue: tiles 0, 1, 2 requested in full; tile 0 landed and published
#pragma unroll
        for (int i = 0; i < NR - 1; ++i) { produce_half(0); produce_half(1); }
        if (total > NR - 2) RG_WAIT(NR - 2); else RG_WAIT(0);
        RG_BARRIER();
        int gt = 0;
        int loc = idx, m0, n0;
        tile_origin(x_first + loc, m0, n0);
        for (;;) {
            if (grp1) RG_BARRIER();                                 // one item behind the first half
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            for (int kt0 = 0; kt0 < nk; kt0 += NR) {
#pragma unroll
                for (int u = 0; u < NR; ++u) {                      // K tile kt0 + u lives in slot u
                    // L0: fragments of k-step 0 (published by the L1 barriers of tile gt - 1 / the prologue)
                    fragload_asm(u, 0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    frag_pin();
                    RG_BARRIER();
                    // M0 (+ first half of the requests of tile gt + 3 into the slot tile gt - 1 left two barriers ago)
                    RG_SB(); m_item(0); RG_SB();
                    RG_BARRIER();
                    // L1: fragments of k-step 1; tile gt + 1 must have landed before the next tile's L0 of EITHER half: every wave waits for its own
                    // pieces here, and both halves' L1 barriers lie before the first half's next L0
                    fragload_asm(u, 1);
                    if (gt + NR - 1 >= total) RG_WAIT(0); else RG_WAIT_L1();
                    frag_pin();
                    RG_BARRIER();
                    // M1 (+ second half of the requests)
                    RG_SB(); m_item(1); RG_SB();
                    RG_BARRIER();
                    ++gt;
                }
            }
            if (!grp1) RG_BARRIER();                                // the halves meet again: the epilogue runs on both together
            gl_epilogue_direct<WGM, WGN, TM, TN, EPI>(g, acc, m0, n0);
            loc += per_xcd;
            if (loc >= x_count) break;
            tile_origin(x_first + loc, m0, n0);
        }
#undef RG_WAIT_L1
        return;
    }
    // prologue: the first NR tiles of the stream are requested; tile 0 must have landed before the first fragment read
#pragma unroll
    for (int i = 0; i < NR; ++i) produce();
    if (total > NR - 1) RG_WAIT(NR - 1); else RG_WAIT(0);
    RG_BARRIER();
    fragload(0, 0, 0);
    int gt = 0;                                                     // K tiles of the stream consumed so far
    bool stores_pending = false;                                    // an epilogue's stores may still be in flight (shared counter: the next wait is vmcnt(0))
    int loc = idx, m0, n0;
    tile_origin(x_first + loc, m0, n0);
    for (;;) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        for (int kt0 = 0; kt0 < nk; kt0 += NR) {
#pragma unroll
            for (int u = 0; u < NR; ++u) {                          // K tile kt0 + u lives in slot u
                fragload(u, 1, 1);
                RG_SB(); mma(0); RG_SB();
                // tile gt + 1 must have landed: the tiles gt + 2, gt + 3 requested after it may stay in flight -- unless they do not exist
                // (end of the stream) or stores were issued since (one counter, out-of-order completion between loads and stores)
                if (stores_pending || gt + NR - 1 >= total) { RG_WAIT(0); stores_pending = false; }
                else RG_WAIT(NR - 2);
                RG_BARRIER();                                       // tile gt + 1 is in LDS for every wave; every wave holds its last fragments of tile gt
                if (!late) produce();                               // tile gt + NR into the slot tile gt just left
                if (gt + 1 < total) fragload((u + 1) & (NR - 1), 0, 0);
                RG_SB(); mma(1); RG_SB();
                if constexpr (STAGGER) { if (late) produce(); }     // (same order of requests per wave: the counted waits are unchanged)
                ++gt;
            }
        }
        gl_epilogue_direct<WGM, WGN, TM, TN, EPI>(g, acc, m0, n0);
        stores_pending = true;
        loc += per_xcd;
        if (loc >= x_count) break;
        tile_origin(x_first + loc, m0, n0);
    }
#undef RG_SB
#undef RG_WAIT
#undef RG_WAIT_N
#undef RG_BARRIER
}

// applies: persistent walk over more tiles than CUs, K a multiple of 4 x 32, the direct epilogue's output form
template <int EPI>
static bool gemm_bf16_ring_applies(const GemmArgs &a) {
    return EPI != EPI_RESID && !a.out_blocked && !a.a_blocked && a.sigma_cols == 0 && a.remap_rows == 0 && (a.N % 16) == 0 && (a.ldo % 8) == 0 && (a.K % 128) == 0 && (a.lda % 8) == 0 &&
           (a.ldw % 8) == 0;
}

// mode: 0 = the plain ring, 1 = STAGGER, 2 = PHASED, 3 = PHASED + s_setprio
template <int WGM, int WGN, int TM, int TN, int EPI>
static void launch_gemm_bf16_ring(const GemmArgs &a, hipStream_t s, int mode = 0) {
    if constexpr (EPI != EPI_RESID) {
        constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
        constexpr int NOUT = (EPI == EPI_GLU) ? BN / 2 : BN;
        const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + NOUT - 1) / NOUT;
        const int n_tiles = tiles_m * tiles_n;
        constexpr size_t lds = 4 * (size_t)(BM + BN) * 32 * 2;
        if (mode == 2 || mode == 3) {                               // PHASED (3: + s_setprio)
            if (mode == 3) {
                auto kern = &gemm_bf16_ring_kernel<WGM, WGN, TM, TN, EPI, false, true, true>;
                static DynLdsSlots slots;
                ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
                hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles);
                return;
            }
            auto kern = &gemm_bf16_ring_kernel<WGM, WGN, TM, TN, EPI, false, true, false>;
            static DynLdsSlots slots;
            ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
            hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles);
            return;
        }
        if (mode == 1) {
            auto kern = &gemm_bf16_ring_kernel<WGM, WGN, TM, TN, EPI, true>;
            static DynLdsSlots slots;
            ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
            hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles);
            return;
        }
        auto kern = &gemm_bf16_ring_kernel<WGM, WGN, TM, TN, EPI>;
        static DynLdsSlots slots;
        ensure_dyn_lds(slots, reinterpret_cast<const void *>(kern), lds);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WGM * WGN), lds, s, a, tiles_n, n_tiles);
    }
}

}  // namespace pk
#endif  // PK_GEMM_BF16_RING_HPP
